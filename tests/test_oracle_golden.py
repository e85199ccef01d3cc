"""Pins oracle/ (the CPU restatement) to golden vectors captured from the imported
reference (tests/golden/gen_golden.py).  CPU only."""
import torch
import pytest

from conftest import load_golden
from oracle import (causal_conv1d_oracle, channel_forward_oracle, fastvim_forward_oracle, fastvim_mixer_oracle,
                    fused_add_norm_oracle, make_state_dict, selective_scan_oracle,
                    selective_scan_ref_port)
from oracle.scan import compressed_scan_oracle

F64 = torch.float64


def close(a, b, rtol, atol):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err:.3e} (max |ref| {b.abs().max():.3e})"


@pytest.mark.parametrize("case", sorted(load_golden("scan.pt").keys()))
def test_scan_oracle_vs_reference(case):
    c = load_golden("scan.pt")[case]
    inp = c["inputs"]
    leaves = {k: v.clone().requires_grad_() for k, v in inp.items() if v is not None}
    out, last = selective_scan_oracle(
        leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves.get("D"),
        leaves.get("z"), leaves["delta_bias"], c["softplus"], True, compute_dtype=F64, out_dtype=F64)
    # reference runs fp32 (selective_scan_interface.py:152-153): agreement at fp32 rounding level
    scale = max(1.0, c["out"].abs().max().item())
    close(out, c["out"], 0, 2e-5 * scale)
    close(last, c["last_state"], 0, 2e-5 * max(1.0, c["last_state"].abs().max().item()))
    out.backward(c["g"].double())
    for k, gref in c["grads"].items():
        gs = max(1.0, gref.abs().max().item())
        close(leaves[k].grad, gref, 0, 1e-4 * gs)
    # fp32 port (the timed CPU baseline) must agree with the reference to fp32 rounding
    outp, lastp = selective_scan_ref_port(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"],
                                          inp["D"], inp["z"], inp["delta_bias"], c["softplus"], True)
    close(outp, c["out"], 0, 2e-5 * scale)
    # bf16 I/O: reference rounds once at the end -> oracle rounded to bf16 within 1 bf16 ulp
    bf = {k: (v.bfloat16() if k in ("u", "delta", "B", "C", "z") and v is not None else v)
          for k, v in inp.items()}
    ob = selective_scan_oracle(bf["u"], bf["delta"], bf["A"], bf["B"], bf["C"], bf["D"], bf["z"],
                               bf["delta_bias"], c["softplus"], compute_dtype=F64)
    assert ob.dtype == torch.bfloat16
    ulp = (c["out_bf16"].float().abs() * 2.0 ** -7).clamp_min(1e-6)
    assert ((ob.float() - c["out_bf16"].float()).abs() <= ulp).all()


@pytest.mark.parametrize("case", sorted(load_golden("compressed_scan.pt").keys()))
def test_compressed_scan_oracle(case):
    c = load_golden("compressed_scan.pt")[case]
    i = c["inputs"]
    out, last = compressed_scan_oracle(i["u_full"], i["u_c"], i["delta"], i["A"], i["B"], i["C"], i["D"],
                                       i["delta_bias"], False, True, compute_dtype=F64, out_dtype=F64)
    close(out, c["out"], 0, 2e-5 * max(1.0, c["out"].abs().max().item()))
    close(last, c["last_state"], 0, 2e-5 * max(1.0, c["last_state"].abs().max().item()))


@pytest.mark.parametrize("case", sorted(load_golden("conv1d.pt").keys()))
def test_conv_oracle(case):
    c = load_golden("conv1d.pt")[case]
    x = c["x"].clone().requires_grad_()
    w = c["w"].clone().requires_grad_()
    b = c["b"].clone().requires_grad_() if c["b"] is not None else None
    y = causal_conv1d_oracle(x, w, b, c["act"], compute_dtype=F64, out_dtype=F64)
    close(y, c["y"], 1e-5, 1e-5)
    y.backward(c["g"].double())
    close(x.grad, c["dx"], 1e-5, 1e-5)
    close(w.grad, c["dw"], 1e-5, 2e-5)
    if b is not None:
        close(b.grad, c["db"], 1e-5, 2e-5)
    # anti-causal form == flip(causal(flip(x)))  (mamba_simple_faster.py:272,280-285)
    ya = causal_conv1d_oracle(c["x"], c["w"], c["b"], c["act"], anticausal=True, compute_dtype=F64, out_dtype=F64)
    yf = causal_conv1d_oracle(c["x"].flip(-1), c["w"], c["b"], c["act"], compute_dtype=F64, out_dtype=F64).flip(-1)
    close(ya, yf, 0, 1e-12)


@pytest.mark.parametrize("case", sorted(load_golden("norm.pt").keys()))
def test_norm_oracle(case):
    c = load_golden("norm.pt")[case]
    outs = fused_add_norm_oracle(c["x"], c["w"], c["b"], c["residual"], c["eps"], c["prenorm"], True, c["rms"])
    ref = c["out"]
    if not c["prenorm"]:
        outs, ref = (outs,), (ref,)
    for o, r in zip(outs, ref):
        assert o.dtype == r.dtype
        tol = 2e-2 if r.dtype == torch.bfloat16 else 2e-6
        close(o, r, tol, tol)


@pytest.mark.parametrize("case", ["d32_4x4", "d32_3x5", "d192_14x14"])
def test_mixer_oracle(case):
    c = load_golden("mixer.pt")[case]
    if "state_dict" in c:
        sd = c["state_dict"]
    else:
        r = c["param_recipe"]
        full = make_state_dict(seed=r["seed"], embed_dim=r["embed_dim"], depth=r["depth"])
        sd = {k[len(r["prefix"]):]: v for k, v in full.items() if k.startswith(r["prefix"])}
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    h = c["hidden"].clone().requires_grad_()
    y = fastvim_mixer_oracle(p, h, c["token_size"], compute_dtype=F64, out_dtype=F64)
    s = max(1.0, c["out"].abs().max().item())
    close(y, c["out"], 0, 2e-5 * s)
    y.backward(c["g"].double())
    close(h.grad, c["dhidden"], 0, 1e-4 * max(1.0, c["dhidden"].abs().max().item()))
    for k, gref in c["grads"].items():
        close(p[k].grad, gref, 0, 2e-4 * max(1.0, gref.abs().max().item()))


@pytest.mark.parametrize("case", ["tiny_64x64", "tiny_48x80"])
def test_model_oracle_tiny(case):
    c = load_golden("model_tiny.pt")[case]
    sd = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    logits, hiddens = fastvim_forward_oracle(sd, c["x"], patch_size=16, depth=4, compute_dtype=F64,
                                             return_hidden=True)
    close(logits, c["logits"], 0, 5e-5 * max(1.0, c["logits"].abs().max().item()))
    for h, href in zip(hiddens, c["hiddens"]):
        close(h, href, 0, 5e-5 * max(1.0, href.abs().max().item()))
    logits.backward(c["g"].double())
    for k, gref in c["grads"].items():
        close(sd[k].grad, gref, 0, 5e-4 * max(1.0, gref.abs().max().item()))


def test_model_oracle_fastvim_t_logits():
    """BASELINE config 1: FastVim-T 224x224 bs=2 through the pure-PyTorch path on CPU."""
    c = load_golden("model_fastvim_t.pt")
    sd = make_state_dict(seed=c["param_seed"], embed_dim=192, depth=24)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(c["x_seed"]))
    assert torch.equal(x[0, 0, :2, :8], c["x_probe"]), "seeded input recipe drifted"
    with torch.no_grad():
        logits = fastvim_forward_oracle(sd, x, compute_dtype=torch.float32)
    close(logits, c["logits"], 0, 2e-3 * max(1.0, c["logits"].abs().max().item()))


def test_model_oracle_fastvim_b_logits():
    """BASELINE config 3 model (FastVim-B 224x224) at bs=2 through the pure-PyTorch path on CPU, fp32."""
    c = load_golden("model_fastvim_b.pt")
    sd = make_state_dict(seed=c["param_seed"], embed_dim=768, depth=24)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(c["x_seed"]))
    assert torch.equal(x[0, 0, :2, :8], c["x_probe"]), "seeded input recipe drifted"
    with torch.no_grad():
        logits = fastvim_forward_oracle(sd, x, compute_dtype=torch.float32)
    close(logits, c["logits"], 0, 2e-3 * max(1.0, c["logits"].abs().max().item()))


def test_model_oracle_colwise():
    """scanpath_type="colwise" (Pool_row, models/fastvim.py:45-51, 97-98): logits and gradients."""
    c = load_golden("model_colwise.pt")
    sd = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    logits = fastvim_forward_oracle(sd, c["x"], patch_size=16, depth=4, compute_dtype=F64, scanpath_type="colwise")
    close(logits, c["logits"], 0, 5e-5 * max(1.0, c["logits"].abs().max().item()))
    logits.backward(c["g"].double())
    for k, gref in c["grads"].items():
        close(sd[k].grad, gref, 0, 5e-4 * max(1.0, gref.abs().max().item()))
    # and the path is not the rowwise one
    with torch.no_grad():
        rw = fastvim_forward_oracle({k: v.detach() for k, v in sd.items()}, c["x"], patch_size=16, depth=4, compute_dtype=F64)
    assert (rw - c["logits"]).abs().max().item() > 1e-3


# ---- FastChannelVim (channel-wise tokenization, Channel-First): oracle vs the imported reference
@pytest.mark.parametrize("case", ["mixer_d32_4x4_t3", "mixer_d32_2x6_t5", "mixer_d64_4x2_t8"])
def test_channel_mixer_oracle(case):
    c = load_golden("channel.pt")[case]
    p = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    h = c["hidden"].clone().requires_grad_()
    y = fastvim_mixer_oracle(p, h, c["token_size"], tokens_per_patch=c["tokens_per_patch"],
                             compute_dtype=F64, out_dtype=F64)
    close(y, c["out"], 0, 2e-5 * max(1.0, c["out"].abs().max().item()))
    y.backward(c["g"].double())
    close(h.grad, c["dhidden"], 0, 1e-4 * max(1.0, c["dhidden"].abs().max().item()))
    for k, gref in c["grads"].items():
        close(p[k].grad, gref, 0, 2e-4 * max(1.0, gref.abs().max().item()))


@pytest.mark.parametrize("case", ["tiny_64x64_c3", "tiny_64x96_c5_hcs"])
def test_channel_model_oracle(case):
    from oracle import channel_forward_oracle
    c = load_golden("channel.pt")[case]
    sd = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    logits = channel_forward_oracle(sd, c["x"], patch_size=16, depth=4, channels=c["subset"], compute_dtype=F64)
    close(logits, c["logits"], 0, 5e-5 * max(1.0, c["logits"].abs().max().item()))
    logits.backward(c["g"].double())
    for k, gref in c["grads"].items():
        close(sd[k].grad, gref, 0, 5e-4 * max(1.0, gref.abs().max().item()))


@pytest.mark.parametrize("case", ["spatial_first_64x96_c3", "compress2d_64x96_c4", "compress2d_64x64_c3_nopos"])
def test_channel_variant_oracle(case):
    """scan_order="Spatial-First" and the 2-D compress model against the imported reference."""
    c = load_golden("channel_variants.pt")[case]
    sd = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    logits = channel_forward_oracle(sd, c["x"], patch_size=16, depth=c["depth"], compute_dtype=F64,
                                    scan_order=c["kw"].get("scan_order", "Channel-First"),
                                    compress2d=case.startswith("compress2d"))
    close(logits, c["logits"], 0, 5e-5 * max(1.0, c["logits"].abs().max().item()))
    logits.backward(c["g"].double())
    for k, gref in c["grads"].items():
        close(sd[k].grad, gref, 0, 5e-4 * max(1.0, gref.abs().max().item()))


# ---- Vim baseline (un-pooled bidirectional mixer, middle class token): oracle vs the imported reference
@pytest.mark.parametrize("case", ["mixer_d32_L9", "mixer_d32_L13", "mixer_d64_L20"])
def test_vim_mixer_oracle(case):
    from oracle import vim_mixer_oracle
    c = load_golden("vim.pt")[case]
    p = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    h = c["hidden"].clone().requires_grad_()
    y = vim_mixer_oracle(p, h, compute_dtype=F64, out_dtype=F64)
    close(y, c["out"], 0, 2e-5 * max(1.0, c["out"].abs().max().item()))
    y.backward(c["g"].double())
    close(h.grad, c["dhidden"], 0, 1e-4 * max(1.0, c["dhidden"].abs().max().item()))
    for k, gref in c["grads"].items():
        close(p[k].grad, gref, 0, 2e-4 * max(1.0, gref.abs().max().item()))


def test_vim_model_oracle():
    from oracle import vim_forward_oracle
    c = load_golden("vim.pt")["tiny_64x64_cls"]
    sd = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    logits = vim_forward_oracle(sd, c["x"], patch_size=16, depth=4, compute_dtype=F64)
    close(logits, c["logits"], 0, 5e-5 * max(1.0, c["logits"].abs().max().item()))
    logits.backward(c["g"].double())
    for k, gref in c["grads"].items():
        close(sd[k].grad, gref, 0, 5e-4 * max(1.0, gref.abs().max().item()))


@pytest.mark.parametrize("case", ["d32_4x4_keep6", "d32_3x5_keep9", "d32_4x4_keep7_unsorted", "d64_6x6_keep9"])
def test_masked_mixer_oracle_matches_reference(case):
    """MAE masked mixer: the flip-free oracle against Mamba_masked of the imported reference, forward,
    input gradient and every parameter gradient (sorted and unsorted ids_keep, rows without kept tokens)."""
    from oracle import masked_mixer_oracle
    c = load_golden("masked.pt")[case]
    p = {k: v.clone().requires_grad_() for k, v in c["state_dict"].items()}
    h = c["hidden"].clone().requires_grad_()
    y = masked_mixer_oracle(p, h, c["ids_keep"], c["token_size"], compute_dtype=torch.float64, out_dtype=torch.float64)
    y.backward(c["g"].double())
    assert (y - c["out"].double()).abs().max().item() <= 2e-6 * max(1.0, c["out"].abs().max().item())
    assert (h.grad - c["dhidden"]).abs().max().item() <= 5e-6 * max(1.0, c["dhidden"].abs().max().item())
    for k, g in c["grads"].items():
        assert (p[k].grad - g).abs().max().item() <= 1e-5 * max(1.0, g.abs().max().item()), k


@pytest.mark.parametrize("case", ["tiny_64_keep4", "tiny_96_keep9"])
def test_mae_oracle_matches_reference(case):
    """FastVim MAE pre-training step: the oracle against MaskedAutoencoderViM of the imported
    reference -- loss, per-patch prediction, mask and parameter gradients, with the masking noise captured."""
    from oracle import mae_forward_oracle
    c = load_golden("mae.pt")[case]
    cfg = c["cfg"]
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in c["state_dict"].items()}
    loss, pred, mask = mae_forward_oracle(sd, c["x"], c["noise"], patch_size=cfg["patch_size"], depth=cfg["depth"],
                                          decoder_depth=cfg["decoder_depth"])
    loss.backward()
    assert abs(loss.item() - c["loss"].item()) <= 1e-6 * max(1.0, abs(c["loss"].item()))
    assert (pred - c["pred"].double()).abs().max().item() <= 5e-6 * max(1.0, c["pred"].abs().max().item())
    assert torch.equal(mask.float(), c["mask"].float())
    for k, g in c["grads"].items():
        assert (sd[k].grad - g).abs().max().item() <= 1e-5 * max(1.0, g.abs().max().item()), k


def test_soft_target_ce_oracle_known_answers():
    """oracle.soft_target_ce_oracle (timm.loss.SoftTargetCrossEntropy's published formula; timm is not vendored in the
    reference) on cases with closed-form answers: uniform logits give log C for any target summing to 1; a one-hot
    target reduces to ordinary cross-entropy, whose gradient is (softmax - onehot) / B."""
    import math
    import torch.nn.functional as F
    from oracle import soft_target_ce_oracle
    t = torch.tensor([[0.25, 0.25, 0.5], [1.0, 0.0, 0.0]])          # exactly representable, rows sum to 1
    loss, g = soft_target_ce_oracle(torch.zeros(2, 3), t)
    assert abs(loss.item() - math.log(3.0)) < 1e-12
    assert torch.allclose(g, (torch.full((2, 3), 1.0 / 3.0).double() - t.double()) / 2)
    torch.manual_seed(0)
    x = torch.randn(4, 7)
    y = torch.tensor([1, 0, 6, 3])
    loss, g = soft_target_ce_oracle(x, F.one_hot(y, 7).float())
    assert abs(loss.item() - F.cross_entropy(x.double(), y).item()) < 1e-12
    assert torch.allclose(g, (F.softmax(x.double(), -1) - F.one_hot(y, 7).double()) / 4)
