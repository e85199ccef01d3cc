"""csrc/glue.hip against the library expressions it replaces (models/fastvim.py:95-101, 175-178, 500, 529-531)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("shape", [(4, 3, 224, 224, 16), (2, 3, 64, 96, 16), (2, 1, 32, 48, 8), (1, 3, 512, 512, 16)])
@pytest.mark.parametrize("idt,odt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                     (torch.bfloat16, torch.bfloat16)])
def test_patch_unfold_is_the_strided_copy(shape, idt, odt):
    from fastvim_amd import glue_ops as G
    B, C, H, W, P = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, C, H, W, device=_dev(), generator=g).to(idt)
    assert G.patch_unfold_ok(x, P, P)
    got = G.patch_unfold(x, P, P, odt)
    gh, gw = H // P, W // P
    ref = x.reshape(B, C, gh, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * P * P).to(odt)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,Ltok,K,D,has_bias,has_pos", [(8, 196, 768, 192, True, True), (3, 24, 768, 384, True, False),
                                                        (2, 196, 768, 192, False, True)])
def test_patch_projection_epilogue_bitwise_and_gradients(B, Ltok, K, D, has_bias, has_pos):
    """One GEMM with the per-token table in its epilogue == LinearFn followed by _EmbedEpilogueFn, values and gradients."""
    from fastvim_amd.fastvim import _EmbedEpilogueFn, _PatchProjFn
    from fastvim_amd.mamba_simple_faster import LinearFn
    g = torch.Generator(device="cuda").manual_seed(2)
    rn = lambda *s: torch.randn(*s, device=_dev(), generator=g)
    patches = rn(B, Ltok, K).bfloat16()
    gout = rn(B, Ltok, D)
    outs = []
    for fused in (True, False):
        W = (rn(D, K) * K ** -0.5).requires_grad_()
        W.data.copy_((torch.arange(D * K, device=_dev()).view(D, K) % 17 - 8).float() * 0.01)
        bias = (torch.linspace(-1, 1, D, device=_dev())).requires_grad_() if has_bias else None
        pos = (torch.sin(torch.arange(Ltok * D, device=_dev()).float()).view(1, Ltok, D)).requires_grad_() if has_pos else None
        if fused:
            y = _PatchProjFn.apply(patches, W, bias, pos, torch.bfloat16)
        else:
            y = _EmbedEpilogueFn.apply(LinearFn.apply(patches, W, torch.bfloat16), bias, pos)
        y.backward(gout)
        outs.append((y.detach(), W.grad, None if bias is None else bias.grad, None if pos is None else pos.grad))
    for a, b in zip(*outs):
        if a is None:
            assert b is None
        else:
            assert a.dtype == b.dtype and torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Ltok,D", [(128, 196, 192), (8, 1024, 768), (3, 7, 20)])
def test_mean_pool(dtype, B, Ltok, D):
    from fastvim_amd.glue_ops import MeanPoolFn
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, Ltok, D, device=_dev(), generator=g).to(dtype).requires_grad_()
    y = MeanPoolFn.apply(x)
    ref = x.detach().double().mean(1)
    tol = 2 ** -8 if dtype == torch.bfloat16 else 1e-6              # one rounding of the fp32 mean to the storage type
    assert y.dtype == dtype and torch.allclose(y.double(), ref, rtol=tol, atol=tol * ref.abs().max().item())
    gy = torch.randn(B, D, device=_dev(), generator=g).to(dtype)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    xr.mean(dim=1).backward(gy)
    assert torch.equal(x.grad, xr.grad)                             # the library's expand / div, bit for bit


def test_droppath_table_matches_expression():
    from fastvim_amd.glue_ops import droppath_table_
    g = torch.Generator(device="cuda").manual_seed(4)
    mods, batch = 25, 128
    u = torch.rand(mods, batch, device=_dev(), generator=g)
    p = torch.linspace(0.0, 0.3, mods, device=_dev())
    keep, inv = (1 - p)[:, None].contiguous(), (1 / (1 - p))[:, None].contiguous()
    ref = u.clone().add_(keep).floor_().mul_(inv)
    got = droppath_table_(u.clone(), keep, inv)
    assert torch.equal(got, ref)
    assert 0 < (got == 0).float().mean() < 0.4


def test_scale_cast_and_column_sum():
    from fastvim_amd.glue_ops import column_sum, scale_cast
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(128, 1000, device=_dev(), generator=g)
    s = torch.tensor(0.37, device=_dev())
    assert torch.equal(scale_cast(x, s, torch.bfloat16), (x * s).bfloat16())
    assert torch.equal(scale_cast(x, s, torch.float32), x * s)
    xb = x.bfloat16()
    ref = xb.double().sum(0)
    got = column_sum(xb)
    assert torch.allclose(got.double(), ref, rtol=1e-5, atol=1e-4)
    acc = torch.ones(1000, device=_dev())
    column_sum(x, out=acc, accumulate=True)
    assert torch.allclose(acc.double(), 1 + x.double().sum(0), rtol=1e-5, atol=1e-4)


def test_loss_backward_uses_fused_scale_cast():
    from fastvim_amd.losses import SoftTargetCrossEntropy
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(16, 1000, device=_dev(), generator=g).bfloat16().requires_grad_()
    t = torch.softmax(torch.randn(16, 1000, device=_dev(), generator=g), -1)
    loss = SoftTargetCrossEntropy()(x, t)
    (loss * 3.0).backward()
    xr = x.detach().float().requires_grad_()
    (torch.sum(-t * F.log_softmax(xr, dim=-1), dim=-1).mean() * 3.0).backward()
    assert x.grad.dtype == torch.bfloat16
    assert torch.allclose(x.grad.float(), xr.grad, rtol=2 ** -7, atol=1e-6)
