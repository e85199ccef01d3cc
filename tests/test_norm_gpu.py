"""GPU parity of the fused add + RMSNorm/LayerNorm HIP op (rms_norm_fn / layer_norm_fn)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("case", sorted(load_golden("norm.pt").keys()))
def test_norm_vs_reference_golden(case):
    from fastvim_amd.layernorm import layer_norm_fn
    c = load_golden("norm.pt")[case]
    g = lambda t: None if t is None else t.cuda()
    outs = layer_norm_fn(g(c["x"]), g(c["w"]), g(c["b"]), residual=g(c["residual"]), eps=c["eps"],
                         prenorm=c["prenorm"], residual_in_fp32=True, is_rms_norm=c["rms"])
    ref = c["out"]
    if not c["prenorm"]:
        outs, ref = (outs,), (ref,)
    for o, r in zip(outs, ref):
        assert o.dtype == r.dtype and o.shape == r.shape
        tol = 2.0 ** -7 if r.dtype == torch.bfloat16 else 2e-6
        assert _err(o, r) <= tol * max(1.0, r.abs().max().item()), _err(o, r)


@pytest.mark.parametrize("rms", [True, False])
@pytest.mark.parametrize("N", [32, 192, 384, 768, 1280])
@pytest.mark.parametrize("xdt", [torch.float32, torch.bfloat16])
def test_norm_fwd_bwd_vs_oracle(rms, N, xdt):
    from fastvim_amd.layernorm import layer_norm_fn
    from oracle import fused_add_norm_oracle
    torch.manual_seed(N)
    Bsz, Ltok = 3, 37
    x = torch.randn(Bsz, Ltok, N).to(xdt)
    res = torch.randn(Bsz, Ltok, N)
    w = 1 + 0.1 * torch.randn(N)
    b = None if rms else 0.1 * torch.randn(N)
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9])
    xc, rc_, wc = x.clone().requires_grad_(), res.clone().requires_grad_(), w.clone().requires_grad_()
    bc = b.clone().requires_grad_() if b is not None else None
    yo, ro = fused_add_norm_oracle(xc, wc, bc, rc_, 1e-5, True, True, rms, row_scale=scale, compute_dtype=F64)
    xg, rg, wg = x.cuda().requires_grad_(), res.cuda().requires_grad_(), w.cuda().requires_grad_()
    bg = b.cuda().requires_grad_() if b is not None else None
    y, r = layer_norm_fn(xg, wg, bg, residual=rg, eps=1e-5, prenorm=True, residual_in_fp32=True,
                         is_rms_norm=rms, row_scale=scale.cuda())
    assert y.dtype == xdt and r.dtype == torch.float32
    tol = 2.0 ** -7 if xdt == torch.bfloat16 else 3e-6
    assert _err(y, yo) <= tol * max(1.0, yo.abs().max().item())
    assert _err(r, ro) <= 1e-6 * max(1.0, ro.abs().max().item())
    gy, gr = torch.randn(Bsz, Ltok, N), torch.randn(Bsz, Ltok, N)
    (yo.double() * gy.double()).sum().add((ro.double() * gr.double()).sum()).backward()
    torch.autograd.backward((y, r), (gy.cuda().to(xdt), gr.cuda()))
    gtol = 2e-2 if xdt == torch.bfloat16 else 2e-5
    assert _err(xg.grad, xc.grad) <= gtol * max(1.0, xc.grad.abs().max().item())
    assert _err(rg.grad, rc_.grad) <= gtol * max(1.0, rc_.grad.abs().max().item())
    assert _err(wg.grad, wc.grad) <= gtol * max(1.0, wc.grad.abs().max().item())
    if b is not None:
        assert _err(bg.grad, bc.grad) <= gtol * max(1.0, bc.grad.abs().max().item())


def test_norm_full_size_idempotent_scale():
    """config-2 shape (25088 x 192): RMSNorm output is invariant to a positive rescale of the input row."""
    from fastvim_amd.layernorm import rms_norm_fn
    torch.manual_seed(0)
    x = torch.randn(128, 196, 192, device="cuda")
    w = torch.ones(192, device="cuda")
    y1 = rms_norm_fn(x, w, None, eps=0.0)
    y2 = rms_norm_fn(3.0 * x, w, None, eps=0.0)
    assert _err(y1, y2) <= 1e-5
    assert abs(y1.square().mean(-1).sqrt().mean().item() - 1.0) < 1e-4
