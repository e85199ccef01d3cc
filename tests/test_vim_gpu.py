"""GPU parity of the Vim baseline surface (models/vim.py + mamba_simple.py): the
un-pooled bidirectional mixer and the middle-class-token backbone, on the fused HIP kernels (rows x 1 x t grid),
against golden vectors captured from the imported reference and against the fp64 oracle."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("case", ["mixer_d32_L9", "mixer_d32_L13", "mixer_d64_L20"])
def test_vim_mixer_fp32_vs_reference_golden(case):
    from fastvim_amd.mamba_simple import Mamba
    c = load_golden("vim.pt")[case]
    sd = c["state_dict"]
    m = Mamba(sd["in_proj.weight"].shape[1]).cuda()
    m.load_state_dict(sd, strict=True)
    hg = c["hidden"].cuda().requires_grad_()
    y = m(hg)
    assert _err(y, c["out"]) <= 1e-5 * max(1.0, c["out"].abs().max().item()), _err(y, c["out"])
    y.backward(c["g"].cuda())
    assert _err(hg.grad, c["dhidden"]) <= 2e-5 * max(1.0, c["dhidden"].abs().max().item())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 1e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


@pytest.mark.parametrize("d_model,L,dtype", [(192, 197, torch.float32), (192, 197, torch.bfloat16), (384, 196, torch.float32),
                                             (96, 50, torch.float32)])
def test_vim_mixer_vs_oracle(d_model, L, dtype):
    """Vim-T / Vim-S mixer shapes (197 = 196 patches + class token is prime: a single 197-token row)."""
    from fastvim_amd.mamba_simple import Mamba
    from oracle import vim_mixer_oracle
    torch.manual_seed(L)
    m = Mamba(d_model).cuda()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight") or n.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    h, g = torch.randn(2, L, d_model), torch.randn(2, L, d_model)
    bf = dtype == torch.bfloat16
    if bf:
        h, g = h.bfloat16().float(), g.bfloat16().float()
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = vim_mixer_oracle(p, hc, compute_dtype=F64, out_dtype=F64)
    yref.backward(g.double())
    hg = h.cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
        y = m(hg)
    ty, td, tw = (2e-2, 3e-2, 4e-2) if bf else (2e-5, 5e-5, 2e-4)
    assert _err(y, yref) <= ty * max(1.0, yref.abs().max().item()), _err(y, yref)
    y.backward(g.cuda().to(y.dtype))
    assert _err(hg.grad, hc.grad) <= td * max(1.0, hc.grad.abs().max().item())
    for n, q in m.named_parameters():
        e = _err(q.grad, p[n].grad)
        assert e <= tw * max(1.0, p[n].grad.abs().max().item()), (n, e, p[n].grad.abs().max().item())


def test_vim_model_vs_reference_golden():
    from fastvim_amd.vim import VisionMamba
    c = load_golden("vim.pt")["tiny_64x64_cls"]
    m = VisionMamba(img_size=64, patch_size=16, depth=4, embed_dim=32, channels=3, num_classes=10, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                    if_cls_token=True, use_middle_cls_token=True, drop_path_rate=0.0).cuda().eval()
    m.load_state_dict(c["state_dict"], strict=True)
    logits = m(c["x"].cuda())
    assert _err(logits, c["logits"]) <= 2e-5 * max(1.0, c["logits"].abs().max().item()), _err(logits, c["logits"])
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_vim_tiny_full_size_step():
    """Vim-T 224 px: bf16 autocast training step is finite and bitwise reproducible."""
    from fastvim_amd import vim
    torch.manual_seed(0)
    m = vim.vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2(drop_path_rate=0.0).cuda().train()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    outs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.float().square().mean().backward()
        outs.append((y.detach().clone(), m.layers[5].mixer.A_log.grad.clone(), m.cls_token.grad.clone()))
    assert torch.isfinite(outs[0][0]).all() and y.shape == (4, 1000)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_mm_vim_multiscale_features():
    """MM_Vim.forward (models/vim.py:624-636): no class token, (B, C, H, W) maps of the LayerNorm-ed hidden states of
    out_indices, equal to the oracle's un-pooled block stack."""
    from fastvim_amd.vim import MM_Vim
    from oracle.model import _sub, fused_add_norm_oracle, patch_embed_oracle, vim_mixer_oracle
    torch.manual_seed(5)
    m = MM_Vim(img_size=(64, 96), depth=4, embed_dim=32, out_indices=[1, 3], rms_norm=True, fused_add_norm=True,
               residual_in_fp32=True, if_abs_pos_embed=True, drop_path_rate=0.0).cuda().eval()
    assert not hasattr(m, "cls_token") and not hasattr(m, "head") and m.pos_embed.shape == (1, 24, 32)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x = torch.randn(2, 3, 64, 96)
    outs = m(x.cuda())
    assert len(outs) == 2 and outs[0].shape == (2, 32, 4, 6)
    cd = torch.float64
    h, _ = patch_embed_oracle(sd, x, 16, cd)
    h = h + sd["pos_embed"].to(cd)
    residual, hiddens = None, {}
    for i in range(4):
        sdl = _sub(sd, f"layers.{i}.")
        hn, residual = fused_add_norm_oracle(h, sdl["norm.weight"], None, residual, 1e-5, prenorm=True,
                                             residual_in_fp32=True, is_rms_norm=True, compute_dtype=cd)
        h = vim_mixer_oracle(_sub(sdl, "mixer."), hn, compute_dtype=cd)
        hiddens[i] = h
    for k, idx in enumerate((1, 3)):
        ref = torch.nn.functional.layer_norm(hiddens[idx].float(), (32,), sd[f"outnorm_{k}.weight"], sd[f"outnorm_{k}.bias"])
        ref = ref.view(2, 4, 6, 32).permute(0, 3, 1, 2)
        assert _err(outs[k], ref) <= 5e-5 * max(1.0, ref.abs().max().item())
