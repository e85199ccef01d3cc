"""fv_mixer_combine_out_proj_addnorm (round 5: combine as the A-tile producer of the out_proj + add + RMSNorm launch)
BIT FOR BIT against the two launches it replaces, fv_mixer_combine_fwd + fv_gemm_bf16_addnorm -- whose own parity with the oracle and
the reference goldens is pinned by tests/test_mixer_gpu.py / test_chain_gpu.py / test_model_gpu.py -- and, end to end,
the chained backbone with and without the fusion."""
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
        xz=rn(B, L, 2 * d_in).to(dev, torch.bfloat16), skip=rn(B, L, d_in).to(dev, torch.bfloat16),
        yc=rn(2, B, rows, d_in).to(dev), ln_w=(1 + 0.1 * rn(d_in)).to(dev), ln_b=(0.1 * rn(d_in)).to(dev),
        W=(rn(d, d_in) * d_in ** -0.5).to(dev, torch.bfloat16), res=rn(B * L, d).to(dev), nw=(1 + 0.1 * rn(d)).to(dev),
        scale=((torch.rand(B, generator=g) > 0.3).float() / 0.7).to(dev))


@pytest.mark.parametrize("B,rows,cols,transposed", [(8, 14, 14, False), (8, 14, 14, True), (3, 14, 14, False), (3, 14, 14, True),
                                                    (5, 16, 16, True), (2, 7, 9, False), (2, 7, 9, True), (128, 14, 14, True)])
@pytest.mark.parametrize("with_ln,with_scale", [(True, True), (False, False)])
def test_fused_combine_out_proj_addnorm_equals_the_two_launches(B, rows, cols, transposed, with_ln, with_scale):
    import ctypes
    from fastvim_amd import _lib as L_, mixer_ops as M
    t = _inputs(B, rows, cols, seed=B + rows + 3 * int(transposed))
    lw, lb = (t["ln_w"], t["ln_b"]) if with_ln else (None, None)
    sc = t["scale"] if with_scale else None
    Mrows, d = B * rows * cols, 192
    rps = rows * cols
    assert M.combine_out_proj_addnorm_ok(t["xz"], rows, cols, 1, d)
    # reference: the two launches
    g0, mean0, rstd0 = M.combine_fwd(t["xz"], t["skip"], t["yc"], lw, lb, 1e-5, rows, cols, transposed)
    y0 = torch.empty(Mrows, d, device="cuda", dtype=torch.bfloat16)
    ro0 = torch.empty(Mrows, d, device="cuda")
    rs0 = torch.empty(Mrows, device="cuda")
    rc = L_.lib().fv_gemm_bf16_addnorm(L_.ptr(g0), L_.ptr(t["W"]), L_.ptr(t["res"]), L_.ptr(t["nw"]), L_.ptr(sc), L_.i32(rps),
                                       L_.ptr(y0), L_.ptr(ro0), L_.ptr(rs0), L_.i32(Mrows), L_.i32(d), L_.i32(384),
                                       ctypes.c_long(384), ctypes.c_long(384), ctypes.c_float(1e-5), L_.stream_of(g0))
    L_.check(rc, "gemm_bf16_addnorm")
    out = M.combine_buffers(t["xz"], lw)
    for b_ in out:
        if b_ is not None:
            b_.fill_(float("nan"))
    y, ro, rs = M.combine_out_proj_addnorm(t["xz"], t["skip"], t["yc"], lw, lb, 1e-5, rows, cols, transposed, out, t["W"],
                                           t["res"], t["nw"], sc, rps, 1e-5)
    torch.cuda.synchronize()
    g1, mean1, rstd1 = out
    # bit for bit: the gating arithmetic is compiled in source order in both kernels (#pragma clang fp reassociate(off)
    # contract(off)) and 1 / d_inner comes out of the same run-time reciprocal; behind g everything is the same instruction
    # sequence as fv_gemm_bf16_addnorm
    if with_ln:
        assert torch.equal(mean1, mean0) and torch.equal(rstd1, rstd0)
    assert torch.equal(g1, g0), (g1.float() - g0.float()).abs().max().item()
    assert torch.equal(y, y0) and torch.equal(ro, ro0) and torch.equal(rs, rs0)
    assert torch.isfinite(y.float()).all() and torch.isfinite(ro).all()
    # deterministic
    out2 = M.combine_buffers(t["xz"], lw)
    y2, ro2, rs2 = M.combine_out_proj_addnorm(t["xz"], t["skip"], t["yc"], lw, lb, 1e-5, rows, cols, transposed, out2, t["W"],
                                              t["res"], t["nw"], sc, rps, 1e-5)
    assert torch.equal(out2[0], g1) and torch.equal(y2, y) and torch.equal(ro2, ro) and torch.equal(rs2, rs)


def test_chained_backbone_with_and_without_the_fused_combine(monkeypatch):
    """A 4-block FastVim-T-width backbone, training mode with DropPath, bf16, on the flat training state: loss and every
    gradient with the combine inside the out_proj launch against the separate launches -- bit for bit."""
    import fastvim_amd.mamba_simple_faster as msf
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatTrainingState
    res = []
    for fuse in (True, False):
        monkeypatch.setattr(msf, "COMBINE_IN_OUT_PROJ", fuse)
        calls = []
        real = msf.M.combine_out_proj_addnorm
        monkeypatch.setattr(msf.M, "combine_out_proj_addnorm", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        torch.manual_seed(0)
        m = VisionMamba(img_size=224, patch_size=16, embed_dim=192, depth=4, num_classes=10, drop_path_rate=0.1, rms_norm=True,
                        residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean").cuda().train()
        flat = FlatTrainingState(m)
        x = torch.randn(32, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        flat.zero_grad()
        torch.manual_seed(5)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m(x).float().square().mean()
        loss.backward()
        flat.finish_backward()
        monkeypatch.setattr(msf.M, "combine_out_proj_addnorm", real)
        assert len(calls) == (2 if fuse else 0), len(calls)      # blocks 1 and 2 hand their combine to blocks 2 and 3
        res.append((loss.item(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}))
        flat.close()
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


def test_deferred_combine_is_off_under_saved_tensor_hooks():
    """a deferred combine leaves g / mean / rstd UNWRITTEN when they are saved for backward (the next block's
    launch fills them); a saved-tensor hook that copies at save time (save_on_cpu, non-reentrant checkpointing) would copy
    garbage.  Under such hooks the chain runs the combine as its own launch: logits and gradients equal the plain run's."""
    from fastvim_amd.fastvim import VisionMamba
    torch.manual_seed(0)
    m = VisionMamba(img_size=224, depth=4, embed_dim=192, num_classes=20, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda().train()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    res = []
    for hooked in (False, True):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if hooked:
                with torch.autograd.graph.save_on_cpu():
                    y = m(x)
            else:
                y = m(x)
        y.float().square().mean().backward()
        res.append((y.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0])
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n
