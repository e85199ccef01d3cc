"""fv_mixer_conv_pool_bwd_dgrad (round 6: the conv + pool adjoint as the A-tile producer of the in_proj data gradient + norm
adjoint) against the two launches it replaces, fv_mixer_conv_pool_bwd2 + fv_gemm_bf16_dgrad_addnorm_bwd2 -- whose own parity
with the oracle and the reference goldens is pinned by tests/test_mixer_gpu.py / test_chain_gpu.py / test_model_gpu.py:
BIT FOR BIT for every data tensor (the x half of d xz, d hidden, d residual, the previous block's d g), to fp32 rounding
for the parameter-gradient sums (same addends, rows grouped per pooling-row tile instead of per persistent block) -- and,
end to end, a flat-state training step of the chained backbone with and without the fusion."""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, rows, cols, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    d_in, d = 384, 192
    L = rows * cols
    dev = "cuda"
    return dict(
        xz=rn(B, L, 2 * d_in).to(dev, torch.bfloat16), d_o=rn(B, L, d_in).to(dev, torch.bfloat16),
        dxc=rn(2, B, rows, d_in).to(dev), dxc2=rn(2, B, rows, d_in).to(dev, torch.bfloat16),
        dz=rn(B, L, d_in).to(dev, torch.bfloat16),
        cw=(0.5 * rn(d_in, 4)).to(dev), cb=(0.1 * rn(d_in)).to(dev), cwb=(0.5 * rn(d_in, 4)).to(dev), cbb=(0.1 * rn(d_in)).to(dev),
        D=(1 + 0.1 * rn(d_in)).to(dev), Db=(1 + 0.1 * rn(d_in)).to(dev),
        W_in=(rn(2 * d_in, d) * d ** -0.5).to(dev, torch.bfloat16), W_out=(rn(d, d_in) * d_in ** -0.5).to(dev, torch.bfloat16),
        gg=rn(B * L, d).to(dev), r=rn(B * L, d).to(dev), rstd=(0.5 + torch.rand(B * L, generator=g)).to(dev),
        nw=(1 + 0.1 * rn(d)).to(dev), scale=((torch.rand(B, generator=g) > 0.3).float() / 0.7).to(dev))


@pytest.mark.parametrize("B,rows,cols,transposed", [(8, 14, 14, False), (8, 14, 14, True), (3, 14, 14, True), (5, 16, 16, False),
                                                    (5, 16, 16, True), (2, 14, 16, False), (2, 16, 14, True), (128, 14, 14, True)])
@pytest.mark.parametrize("x2,with_scale,with_gg,second", [(True, True, True, True), (False, False, False, False), (True, False, True, False)])
def test_fused_conv_pool_bwd_dgrad_equals_the_two_launches(B, rows, cols, transposed, x2, with_scale, with_gg, second):
    from fastvim_amd import _lib as L_, mixer_ops as M
    t = _inputs(B, rows, cols, seed=B + rows + 3 * int(transposed))
    d_in, d = 384, 192
    Mrows, rps = B * rows * cols, rows * cols
    sc = t["scale"] if with_scale else None
    gg = t["gg"] if with_gg else None
    dxc2 = t["dxc2"] if x2 else None
    W2 = t["W_out"] if second else None
    assert M.conv_pool_bwd_dgrad_ok(t["xz"], rows, cols, 1, d, False)

    def fresh_dxz():
        dxz = torch.full((B, rows * cols, 2 * d_in), float("nan"), device="cuda", dtype=torch.bfloat16)
        dxz[:, :, d_in:] = t["dz"]          # the z half is combine_bwd's output
        return dxz

    # reference: the two launches
    dxz0 = fresh_dxz()
    p0 = M.conv_pool_bwd(t["xz"], t["d_o"], t["dxc"], t["cw"], t["cb"], t["cwb"], t["cbb"], t["D"], t["Db"], dxz0, rows, cols,
                         transposed, False, 1.0, dxc2=dxc2)
    lib = L_.lib()
    nb0 = lib.fv_gemm_bf16_dgrad_addnorm_blocks(L_.i32(Mrows))
    dx0 = torch.empty(Mrows, d, device="cuda", dtype=torch.bfloat16)
    dri0 = torch.empty(Mrows, d, device="cuda")
    pw0 = torch.empty(nb0, d, device="cuda")
    dg0 = torch.empty(Mrows, d_in, device="cuda", dtype=torch.bfloat16) if second else None
    dxz0_2 = dxz0.view(Mrows, 2 * d_in)
    rc = lib.fv_gemm_bf16_dgrad_addnorm_bwd2(
        L_.ptr(dxz0_2), L_.ptr(t["W_in"]), L_.ptr(gg), L_.ptr(t["r"]), L_.ptr(t["rstd"]), L_.ptr(t["nw"]), L_.ptr(sc), L_.i32(rps),
        L_.ptr(dx0), L_.ptr(dri0), L_.ptr(pw0), L_.i32(Mrows), L_.i32(d), L_.i32(2 * d_in), ctypes.c_long(2 * d_in),
        ctypes.c_long(d), L_.ptr(W2), L_.ptr(dg0), L_.i32(d_in if second else 0), ctypes.c_long(d_in), L_.stream_of(dxz0))
    L_.check(rc, "gemm_bf16_dgrad_addnorm_bwd2")
    # the fused launch
    dxz1 = fresh_dxz()
    W_in_t = t["W_in"].t().contiguous()
    p1, dx1, dri1, pw1, nb1, dg1 = M.conv_pool_bwd_dgrad(
        t["xz"], t["d_o"], t["dxc"], dxc2, t["cw"], t["cb"], t["cwb"], t["cbb"], t["D"], t["Db"], dxz1, rows, cols, transposed,
        1.0, W_in_t, gg, t["r"], t["rstd"], t["nw"], sc, rps, W2=W2)
    torch.cuda.synchronize()
    assert torch.isfinite(dxz1.float()).all() and torch.isfinite(dx1.float()).all() and torch.isfinite(dri1).all()
    assert torch.equal(dxz1, dxz0), (dxz1.float() - dxz0.float()).abs().max().item()
    assert torch.equal(dri1, dri0), (dri1 - dri0).abs().max().item()
    assert torch.equal(dx1, dx0), (dx1.float() - dx0.float()).abs().max().item()
    if second:
        assert torch.equal(dg1, dg0), (dg1.float() - dg0.float()).abs().max().item()
    # parameter-gradient sums: same addends, another (fixed) grouping of the rows
    a, b = p1.reshape(-1), p0.reshape(-1)
    assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-6
    w1, w0 = pw1.sum(0), pw0.sum(0)
    assert (w1 - w0).abs().max().item() <= 1e-4 * w0.abs().max().item() + 1e-5
    # run to run: bitwise
    dxz2 = fresh_dxz()
    p2, dx2, dri2, pw2, _, dg2 = M.conv_pool_bwd_dgrad(
        t["xz"], t["d_o"], t["dxc"], dxc2, t["cw"], t["cb"], t["cwb"], t["cbb"], t["D"], t["Db"], dxz2, rows, cols, transposed,
        1.0, W_in_t, gg, t["r"], t["rstd"], t["nw"], sc, rps, W2=W2)
    assert torch.equal(p2, p1) and torch.equal(pw2, pw1) and torch.equal(dx2, dx1) and torch.equal(dxz2, dxz1)


def test_transposed_in_proj_shadow_follows_the_optimizer_and_in_place_writes():
    """The transposed bf16 shadow the fused launch reads is re-made after every fused optimizer step and after any other
    in-place write to in_proj.weight (version counter), like the plain shadow."""
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    from fastvim_amd.mamba_simple_faster import _shadow, _shadow_t
    torch.manual_seed(0)
    m = VisionMamba(img_size=224, depth=2, embed_dim=192, num_classes=10, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, final_pool_type="mean", drop_path_rate=0.0).cuda().train()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    with FlatTrainingState(m) as flat:
        opt = FlatAdamW(flat, m, lr=1e-2, weight_decay=0.05)
        w = m.layers[1].mixer.in_proj.weight
        assert torch.equal(_shadow_t(w, torch.bfloat16), w.detach().to(torch.bfloat16).t())
        flat.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(x).float().square().mean().backward()
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(w._fv_shadow_t, w.detach().to(torch.bfloat16).t())          # refreshed by the step itself
        assert torch.equal(_shadow_t(w, torch.bfloat16), _shadow(w, torch.bfloat16).t())
        with torch.no_grad():
            w.mul_(0.5)
        assert torch.equal(_shadow_t(w, torch.bfloat16), w.detach().to(torch.bfloat16).t())


def test_flat_step_with_and_without_the_fusion():
    """A flat-state FastVim-T (4 blocks) training step, chained blocks: fused conv adjoint + data gradient against the two
    launches -- logits and every data-path gradient identical; conv / norm parameter gradients to fp32 rounding."""
    from fastvim_amd import mamba_simple_faster as msf
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatTrainingState
    torch.manual_seed(0)
    base = VisionMamba(img_size=224, depth=4, embed_dim=192, num_classes=50, rms_norm=True, residual_in_fp32=True,
                       fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.1).cuda().train()
    x = torch.randn(32, 3, 224, 224, device="cuda")
    g = torch.randn(32, 50, device="cuda")
    res = []
    for on in (True, False):
        m = copy.deepcopy(base)
        old = msf.CONV_IN_DGRAD
        msf.CONV_IN_DGRAD = on
        try:
            with FlatTrainingState(m) as flat:
                flat.zero_grad()
                torch.manual_seed(7)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = m(x)
                (y.float() * g).sum().backward()
                flat.finish_backward()
                torch.cuda.synchronize()
                res.append((y.detach().clone(), flat.grad_flat.clone(), dict(flat.offsets), {n: p.numel() for n, p in m.named_parameters()}))
        finally:
            msf.CONV_IN_DGRAD = old
    assert torch.equal(res[0][0], res[1][0])
    offs, numel = res[0][2], res[0][3]
    for n, o in offs.items():
        a, b = res[0][1][o:o + numel[n]], res[1][1][o:o + numel[n]]
        regrouped = any(k in n for k in ("conv1d", "mixer.D", ".norm.weight", "norm_f"))
        if regrouped:
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-7, n
        else:
            assert torch.equal(a, b), n
