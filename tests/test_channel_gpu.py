"""GPU parity of the FastChannelVim path (channel-wise tokenization, Channel-First; BASELINE config 5,
SURVEY.md section 8 row a16): the channel mixer and backbone mirrors, running the fused HIP kernels
with ``tokens_per_patch > 1``, against golden vectors captured from the imported reference
(mamba_simple_channel_faster.py, models_channel_mamba_faster.py) and against the fp64 oracle."""
import random

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def _cell_perm(Bsz, rows, cols, tpp, transposed):
    """sequence order of the mixer -> memory order held by Block on odd layers (cells transposed)."""
    if not transposed:
        return lambda t: t
    return lambda t: t.reshape(Bsz, rows, cols, tpp, -1).transpose(1, 2).reshape(Bsz, rows * cols * tpp, -1)


@pytest.mark.parametrize("case", ["mixer_d32_4x4_t3", "mixer_d32_2x6_t5", "mixer_d64_4x2_t8"])
@pytest.mark.parametrize("transposed", [False, True])
def test_channel_mixer_fp32_vs_reference_golden(case, transposed):
    from fastvim_amd.mamba_simple_channel_faster import Mamba
    c = load_golden("channel.pt")[case]
    rows, cols = c["token_size"]
    tpp = c["tokens_per_patch"]
    sd = c["state_dict"]
    m = Mamba(sd["in_proj.weight"].shape[1], token_size=[rows, cols]).cuda()
    m.load_state_dict(sd, strict=True)
    h = c["hidden"]
    perm = _cell_perm(h.shape[0], rows, cols, tpp, transposed)
    hg = perm(h).contiguous().cuda().requires_grad_()
    y = m(hg, tpp, transposed_grid=transposed)
    ref = perm(c["out"])
    assert _err(y, ref) <= 1e-5 * max(1.0, ref.abs().max().item()), _err(y, ref)
    y.backward(perm(c["g"]).contiguous().cuda())
    dref = perm(c["dhidden"])
    assert _err(hg.grad, dref) <= 2e-5 * max(1.0, dref.abs().max().item()), _err(hg.grad, dref)
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 1e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


@pytest.mark.parametrize("d_model,grid,tpp,dtype", [
    (384, (14, 14), 8, torch.float32),       # FastChannelVim-S/16 mixer at the config-5 shape (L = 1568, Lc = 112)
    (384, (14, 14), 8, torch.bfloat16),
    (192, (4, 6), 3, torch.float32),
    (768, (2, 4), 2, torch.float32),
    (96, (2, 2), 5, torch.float32),          # generic lane mapping (VEC = 1)
    (384, (6, 4), 1, torch.float32),         # tokens_per_patch == 1 must reduce to the FastVim mixer
])
def test_channel_mixer_vs_oracle(d_model, grid, tpp, dtype):
    from fastvim_amd.mamba_simple_channel_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(d_model + tpp)
    m = Mamba(d_model, token_size=list(grid)).cuda()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight") or n.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    rows, cols = grid
    Bsz, Ltok = 2, rows * cols * tpp
    h = torch.randn(Bsz, Ltok, d_model)
    g = torch.randn(Bsz, Ltok, d_model)
    bf = dtype == torch.bfloat16
    if bf:
        h, g = h.bfloat16().float(), g.bfloat16().float()
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = fastvim_mixer_oracle(p, hc, grid, tokens_per_patch=tpp, compute_dtype=F64, out_dtype=F64)
    yref.backward(g.double())
    tol_y, tol_dh, tol_w = (2e-2, 3e-2, 4e-2) if bf else (2e-5, 5e-5, 2e-4)
    for transposed in (False, True):
        m.zero_grad(set_to_none=True)
        perm = _cell_perm(Bsz, rows, cols, tpp, transposed)
        hg = perm(h).contiguous().cuda().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
            y = m(hg, tpp, transposed_grid=transposed)
        assert _err(y, perm(yref)) <= tol_y * max(1.0, yref.abs().max().item()), _err(y, perm(yref))
        y.backward(perm(g).contiguous().cuda().to(y.dtype))
        assert _err(hg.grad, perm(hc.grad)) <= tol_dh * max(1.0, hc.grad.abs().max().item())
        for n, q in m.named_parameters():
            e = _err(q.grad, p[n].grad)
            assert e <= tol_w * max(1.0, p[n].grad.abs().max().item()), (transposed, n, e, p[n].grad.abs().max().item())


@pytest.mark.parametrize("case", ["tiny_64x64_c3", "tiny_64x96_c5_hcs"])
def test_channel_model_vs_reference_golden(case):
    """Eval mode (all channels) and one training-mode step with the HCS subset the reference drew."""
    from fastvim_amd.models_channel_mamba_faster import VisionMamba
    c = load_golden("channel.pt")[case]
    m = VisionMamba(img_size=c["img"], patch_size=16, depth=4, embed_dim=32, channels=c["channels"],
                    num_classes=10, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                    final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda()
    m.load_state_dict(c["state_dict"], strict=True)
    m.train(c["train"])
    random.seed(c["py_seed"])
    logits = m(c["x"].cuda())
    ref = c["logits"]
    assert _err(logits, ref) <= 2e-5 * max(1.0, ref.abs().max().item()), _err(logits, ref)
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_channelvim_small_config5_shape_vs_oracle():
    """FastChannelVim-S/16 width, 8 channels, 224x224 (tokens (B, 1568, 384)); depth cut to 3 so the
    fp64 oracle finishes in seconds.  fp32 logits/gradients vs the oracle, then bf16 autocast close to fp32."""
    from fastvim_amd.models_channel_mamba_faster import VisionMamba
    from oracle import channel_forward_oracle, make_channel_state_dict
    sd = make_channel_state_dict(seed=5, embed_dim=384, depth=3, channels=8, num_classes=16)
    m = VisionMamba(img_size=224, depth=3, embed_dim=384, channels=8, num_classes=16, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, hcs=False, drop_path_rate=0.0).cuda()
    m.load_state_dict(sd, strict=True)
    x = torch.randn(2, 8, 224, 224, generator=torch.Generator().manual_seed(9))
    g = torch.randn(2, 16, generator=torch.Generator().manual_seed(10))
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = channel_forward_oracle(p, x, depth=3, compute_dtype=F64)
    ref.backward(g.double())
    logits = m(x.cuda())
    s = max(1.0, ref.abs().max().item())
    assert _err(logits, ref) <= 5e-5 * s, _err(logits, ref)
    logits.backward(g.cuda())
    for n, q in m.named_parameters():
        gr = p[n].grad
        e = _err(q.grad, gr)
        assert e <= 5e-4 * max(1.0, gr.abs().max().item()), (n, e, gr.abs().max().item())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = m(x.cuda())
    assert _err(lb, ref) <= 5e-2 * s, _err(lb, ref)


def test_channel_mixer_max_pool_vs_oracle():
    """collapse_method="max" with tokens_per_patch > 1 (mamba_simple_channel_faster.py:258-283)."""
    from fastvim_amd.mamba_simple_channel_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(7)
    d_model, grid, tpp = 64, (4, 6), 3
    m = Mamba(d_model, token_size=list(grid), collapse_method="max").cuda()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    h = torch.randn(2, grid[0] * grid[1] * tpp, d_model)
    g = torch.randn_like(h)
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = fastvim_mixer_oracle(p, hc, grid, tokens_per_patch=tpp, collapse_method="max", compute_dtype=F64, out_dtype=F64)
    yref.backward(g.double())
    hg = h.cuda().requires_grad_()
    y = m(hg, tpp)
    assert _err(y, yref) <= 1e-5 * max(1.0, yref.abs().max().item())
    y.backward(g.cuda())
    assert _err(hg.grad, hc.grad) <= 5e-5 * max(1.0, hc.grad.abs().max().item())
    for n, q in m.named_parameters():
        e = _err(q.grad, p[n].grad)
        assert e <= 2e-4 * max(1.0, p[n].grad.abs().max().item()), (n, e)


@pytest.mark.parametrize("case", ["spatial_first_64x96_c3", "compress2d_64x96_c4", "compress2d_64x64_c3_nopos"])
def test_channel_variants_vs_reference_golden(case):
    """SURVEY section 8 row f3 remainder: scan_order="Spatial-First" of the channel model and the 2-D compress model
    (row scan -> column scan -> channel scan), logits and gradients against goldens captured from the imported
    reference (tests/golden/gen_golden.py: gen_channel_variants)."""
    c = load_golden("channel_variants.pt")[case]
    if case.startswith("compress2d"):
        from fastvim_amd.models_channel_mamba_faster_2dcompress import VisionMamba
    else:
        from fastvim_amd.models_channel_mamba_faster import VisionMamba
    m = VisionMamba(img_size=c["img"], patch_size=16, depth=c["depth"], embed_dim=32, channels=c["channels"],
                    num_classes=10, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                    drop_path_rate=0.0, **c["kw"]).cuda().eval()
    m.load_state_dict(c["state_dict"], strict=True)
    logits = m(c["x"].cuda())
    assert _err(logits, c["logits"]) <= 2e-5 * max(1.0, c["logits"].abs().max().item()), _err(logits, c["logits"])
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_compress2d_small_entry_point_bf16_step():
    """FastChannelVim-S/16 2-D compress entry point at the config-5 input shape (8 channels, 224 px), reduced depth:
    one bf16 training step runs on the fused kernels (channel-scan layers: 1 x 196 cells of 8 tokens, scan length 8;
    row / column layers: 14 x 112 grid) and matches the fp64 oracle."""
    from fastvim_amd.models_channel_mamba_faster_2dcompress import VisionMamba
    from oracle import channel_forward_oracle
    torch.manual_seed(0)
    m = VisionMamba(img_size=224, patch_size=16, depth=3, embed_dim=384, channels=8, num_classes=20, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, hcs=False, drop_path_rate=0.0, if_abs_pos_embed=True).cuda().train()
    x = torch.randn(2, 8, 224, 224)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x.cuda())
    logits.float().square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = channel_forward_oracle(sd, x, patch_size=16, depth=3, compute_dtype=F64, compress2d=True)
    rel = (logits.float().cpu().double() - ref).norm() / ref.norm()
    assert rel <= 3e-2, rel
