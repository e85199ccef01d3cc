"""The short backward scan with the x_proj adjoint's data half folded in (fv_mixer_scan_bwd_xproj, round 5) and its
consumer (fv_mixer_conv_pool_bwd2: the pooled gradient as two addends), against

* fp64 math: d u of the scan (selective_scan_oracle autograd) + d x_dbl @ Wx (reference:
  selective_scan_interface.py:679-696, 726-734);
* the unfolded kernels (fv_mixer_scan_bwd + fv_mixer_xproj_bwd2 + fv_mixer_conv_pool_bwd), which compute the same
  function with another summation order: the per-chunk d x_dbl rows and every parameter-gradient partial bit for bit,
  the bf16 rows of the x_proj weight gradient bit for bit, the pooled gradient and d xz to rounding.
"""
import pytest
import torch

from test_config34_gpu import _err, _scan_cl_case

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _case(Bsz, Lc, R, dtype, seed):
    d_in, N = 384, 16
    xc, x_dbl, Wdt, bdt, A_log, dyc = _scan_cl_case(Bsz, Lc, d_in, R, dtype, seed)
    g = torch.Generator().manual_seed(seed + 1)
    Wx = torch.randn(2, R + 2 * N, d_in, generator=g) * d_in ** -0.5
    return xc, x_dbl, Wdt, bdt, A_log, dyc, Wx


@pytest.mark.parametrize("Bsz,Lc,R", [(5, 14, 12), (4, 16, 12), (2, 14, 2), (3, 14, 6), (128, 14, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_bwd_with_folded_xproj_adjoint(Bsz, Lc, R, dtype):
    from fastvim_amd import mixer_ops as M
    from oracle import selective_scan_oracle
    dev, d_in, N = "cuda", 384, 16
    W = R + 2 * N
    xc, x_dbl, Wdt, bdt, A_log, dyc, Wx = _case(Bsz, Lc, R, dtype, seed=31 * Lc + R)
    a = lambda t: t.to(dev)
    xc_d, xd_d = xc.to(dev, dtype), x_dbl.to(dev, dtype)
    args = (xc_d, xd_d, a(Wdt[0]), a(bdt[0]), a(A_log[0]), a(Wdt[1]), a(bdt[1]), a(A_log[1]), a(dyc))
    assert M.scan_bwd_xproj_ok(xc_d, a(Wdt[0]), False, Lc, Lc, 1)
    dxc, dxc2, chunks, pr = M.scan_bwd_xproj(*args, a(Wx[0]), a(Wx[1]))
    assert dxc2.dtype == dtype and chunks.shape == (2, 2, Bsz * Lc, W)
    # deterministic
    r2 = M.scan_bwd_xproj(*args, a(Wx[0]), a(Wx[1]))
    assert all(torch.equal(p, q) for p, q in zip((dxc, dxc2, chunks, pr), r2))
    # the unfolded pair: same chunk partial rows and parameter-gradient sums, bit for bit
    dxc_o, chunks_o, pr_o = M.scan_bwd(*args, keep_chunks=True)
    assert torch.equal(chunks, chunks_o) and torch.equal(pr, pr_o)
    du_scan = dxc_o.clone()
    rows_o = M.xproj_bwd(chunks_o, xc_d, a(Wx[0]), a(Wx[1]), dxc_o, dw=False)      # (publishes bf16 rows in fp32 mode as well)
    rows = torch.empty_like(rows_o)
    M.chunk_rows_bf16([(chunks, rows)])
    assert torch.equal(rows, rows_o)                      # the weight gradient's operand: bit for bit
    tot = dxc + dxc2.float()
    # fp64: d u through the scan + d x_dbl @ Wx with the kernel's own (fp32) d x_dbl
    dxd = chunks.double().sum(0)                          # (2, M, W)
    ref = du_scan.double().view(2, Bsz * Lc, d_in) + torch.einsum("kmw,kwd->kmd", dxd, Wx.double().to(dev))
    s = max(1.0, ref.abs().max().item())
    # fp32 storage: an exact fp32 FMA chain over the W terms.  bf16 storage: the product runs on the bf16 matrix cores
    # from bf16(d x_dbl) and the bf16 shadow weight (what the reference's autocast backward multiplies) -- each term is
    # off by at most 2^-8 relative -- and the other chunk's half of the sum is stored in bf16 once more
    if dtype == torch.bfloat16:
        # (each workgroup rounds ITS chunk's partial rows: the bound is on the sum of the partials' magnitudes)
        absterm = torch.einsum("kmw,kwd->kmd", chunks.double().abs().sum(0), Wx.double().abs().to(dev))
        tol = 2e-5 * s + 2.0 ** -8 * absterm + 2.0 ** -8 * (ref - du_scan.double().view(2, -1, d_in)).abs()
    else:
        tol = torch.full_like(ref, 2e-5 * s)
    e1 = (tot.view(2, -1, d_in).double() - ref).abs()
    assert (e1 <= tol).all(), (e1 / tol).max().item()
    e2 = (tot.double() - dxc_o.double()).abs().view(2, -1, d_in)       # and against the unfolded kernel's total
    assert (e2 <= tol).all(), (e2 / tol).max().item()
    # against fp64 autograd end to end (scan oracle), fp32 only -- bf16 storage of half of one term is covered above
    if dtype == torch.float32 and Bsz <= 8:
        for k in range(2):
            u = xc[k].double().requires_grad_()
            xd = x_dbl[k].view(Bsz, Lc, W).double().requires_grad_()
            delta = xd[..., :R] @ Wdt[k].double().t()
            y = selective_scan_oracle(u.transpose(1, 2), delta.transpose(1, 2), -torch.exp(A_log[k].double()),
                                      xd[..., R:R + N].transpose(1, 2), xd[..., R + N:].transpose(1, 2), None, None,
                                      bdt[k].double(), True, compute_dtype=F64, out_dtype=F64, reverse=bool(k)).transpose(1, 2)
            y.backward(dyc.double())
            full = u.grad + (xd.grad.view(Bsz * Lc, W) @ Wx[k].double()).view(Bsz, Lc, d_in)
            assert _err(tot[k], full) <= 5e-5 * max(1.0, full.abs().max().item()), (k, _err(tot[k], full))


@pytest.mark.parametrize("cols,transposed", [(14, False), (14, True), (16, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_pool_bwd_takes_the_pooled_gradient_as_two_addends(cols, transposed, dtype):
    """fv_mixer_conv_pool_bwd2(dxc, dxc2) == fv_mixer_conv_pool_bwd(dxc + dxc2): d x and the parameter-gradient partial
    sums.  With dxc2 == 0 bit for bit; with a real second addend to the rounding of one fp32 add."""
    from fastvim_amd import mixer_ops as M
    dev, B, rows, d_in = "cuda", 6, cols, 384
    L_tok = rows * cols
    g = torch.Generator().manual_seed(cols + 2 * int(transposed))
    rn = lambda *s: torch.randn(*s, generator=g)
    xz = rn(B, L_tok, 2 * d_in).to(dev, dtype)
    d_o = rn(B, L_tok, d_in).to(dev, dtype)
    dxc = rn(2, B, rows, d_in).to(dev)
    dxc2 = (0.3 * rn(2, B, rows, d_in)).to(dev, dtype)
    cw, cwb = (0.5 * rn(d_in, 4)).to(dev), (0.5 * rn(d_in, 4)).to(dev)
    cb, cbb = (0.1 * rn(d_in)).to(dev), (0.1 * rn(d_in)).to(dev)
    D, Db = rn(d_in).to(dev), rn(d_in).to(dev)

    def run(a, b):
        dxz = torch.zeros_like(xz)
        p = M.conv_pool_bwd(xz, d_o, a, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, transposed, False, 1.0, dxc2=b)
        return dxz[..., :d_in].clone(), p.clone()

    ref = run(dxc, None)
    z = run(dxc, torch.zeros_like(dxc2))
    assert torch.equal(z[0], ref[0]) and torch.equal(z[1], ref[1])
    two = run(dxc, dxc2)
    one = run(dxc + dxc2.float(), None)
    assert torch.equal(two[0], one[0]) and torch.equal(two[1], one[1])      # the same fp32 sum, taken in the kernel
    assert not torch.equal(two[0], ref[0])


def test_training_step_with_and_without_the_fold_agree(monkeypatch):
    """FastVim-T-width blocks on the flat training state: the step with the x_proj adjoint inside the scan backward
    against the step with the separate launch -- every gradient to rounding (the summation order of one 44-term dot
    product per pooled element differs), the x_proj weight gradient bit for bit."""
    import fastvim_amd.mamba_simple_faster as msf
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatTrainingState
    grads = []
    for fold in (True, False):
        monkeypatch.setattr(msf, "XPROJ_IN_SCAN", fold)
        torch.manual_seed(0)
        m = VisionMamba(img_size=224, patch_size=16, embed_dim=192, depth=2, num_classes=10, drop_path_rate=0.0, rms_norm=True,
                        residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean").cuda().train()
        flat = FlatTrainingState(m)
        x = torch.randn(32, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        calls = []
        real = msf.M.scan_bwd_xproj
        monkeypatch.setattr(msf.M, "scan_bwd_xproj", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        flat.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = m(x).float().square().mean()
        loss.backward()
        flat.finish_backward()
        monkeypatch.setattr(msf.M, "scan_bwd_xproj", real)
        assert (len(calls) == 2) == fold
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
        flat.close()
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        s = max(b.abs().max().item(), 1e-6)
        if "x_proj" in n and ".1." in n:       # last block: its d x_dbl rows do not depend on the fold
            assert torch.equal(a, b), n
        assert _err(a, b) <= 2e-2 * s, (n, _err(a, b), s)
