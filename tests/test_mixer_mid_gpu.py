"""The opt-in fused forward launch of the mixer's middle (csrc/mixer_mid_fwd.hip: conv + pool + skip -> x_proj + dt_proj +
scan -> combine, an image carried by a pair of workgroups that hand xc and yc over through memory) against the three
launches it replaces, on both grid orientations, with the hand-off flags left at zero.  Same per-lane arithmetic and
summation orders; -ffast-math contracts the two translation units differently in a few places, so a few elements per
million of the bf16 outputs sit one bf16 ulp apart and what is computed from them follows -- the bounds below are that,
not a numerical tolerance of the method (the mixer's parity with the oracle is tests/test_mixer_gpu.py, both paths)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _opt_in(monkeypatch):
    from fastvim_amd import mixer_ops as M
    monkeypatch.setattr(M, "MID_FWD", True)


def _close(name, a, b):
    """bf16 outputs: a rounding flip of an element (one bf16 ulp, two across a power of two; for x_dbl, a sum over 384
    channels, a flip of the sum at the tensor's scale), in fewer than 1e-3 of the elements; fp32 ones (yc, mean, rstd):
    what those flips do downstream -- 2^-7 of the tensor's scale."""
    assert a.dtype == b.dtype and a.shape == b.shape, name
    af, bf = a.float(), b.float()
    d = (af - bf).abs()
    gmax = bf.abs().max().item()
    if a.dtype == torch.bfloat16:
        bound = torch.maximum(2.0 ** -6 * torch.maximum(af.abs(), bf.abs()), torch.full_like(d, 2.0 ** -8 * gmax))
        assert (d <= bound).all(), (name, d.max().item(), gmax)
        assert (d != 0).float().mean().item() < 1e-3, (name, (d != 0).float().mean().item())
    else:
        assert (d <= 2.0 ** -7 * gmax).all(), (name, d.max().item(), gmax)


def _inputs(B, rows, d_in, R, seed, with_ln=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    N = 16
    L = rows * rows
    t = dict(xz=rn(B, L, 2 * d_in).bfloat16(), cw=rn(d_in, 4) * 0.5, cb=rn(d_in) * 0.1, cwb=rn(d_in, 4) * 0.5, cbb=rn(d_in) * 0.1,
             D=rn(d_in), Db=rn(d_in), Wx2=(rn(2, R + 2 * N, d_in) * d_in ** -0.5).bfloat16(),
             Wdt=rn(d_in, R) * R ** -0.5, bdt=rn(d_in) - 4.0, Wdtb=rn(d_in, R) * R ** -0.5, bdtb=rn(d_in) - 4.0,
             A=torch.log(torch.arange(1, N + 1, device="cuda", dtype=torch.float32)).repeat(d_in, 1).contiguous(),
             Ab=torch.log(torch.arange(1, N + 1, device="cuda", dtype=torch.float32) * 0.7).repeat(d_in, 1).contiguous(),
             lnw=(1 + 0.1 * rn(d_in)) if with_ln else None, lnb=0.1 * rn(d_in) if with_ln else None)
    return t


def _three(t, rows, transposed, scaling=1.0):
    from fastvim_amd import mixer_ops as M
    xc, skip = M.conv_pool_fwd(t["xz"], t["cw"], t["cb"], t["cwb"], t["cbb"], rows, rows, transposed, False, scaling, D=t["D"], D_b=t["Db"])
    x_dbl, yc = M.xproj_scan_fwd(xc, t["Wx2"], t["Wdt"], t["bdt"], t["A"], t["Wdtb"], t["bdtb"], t["Ab"])
    g, mean, rstd = M.combine_fwd(t["xz"], skip, yc, t["lnw"], t["lnb"], 1e-5, rows, rows, transposed)
    return xc, skip, x_dbl, yc, g, mean, rstd


def _one(t, rows, transposed, scaling=1.0):
    from fastvim_amd import mixer_ops as M
    return M.mixer_mid_fwd(t["xz"], t["cw"], t["cb"], t["cwb"], t["cbb"], t["D"], t["Db"], t["Wx2"], t["Wdt"], t["bdt"], t["A"],
                           t["Wdtb"], t["bdtb"], t["Ab"], t["lnw"], t["lnb"], 1e-5, rows, rows, transposed, scaling)


NAMES = ("xc", "skip", "x_dbl", "yc", "g", "mean", "rstd")


@pytest.mark.parametrize("B,rows,R,transposed,with_ln", [
    (128, 14, 12, False, True),      # FastVim-T 224 px, the benchmark shape: one workgroup on every CU
    (128, 14, 12, True, True),       # odd layers: the grid read through swapped strides
    (3, 14, 12, False, True), (1, 14, 12, True, False),
    (5, 16, 12, False, True), (64, 16, 12, True, True),          # 256 px grid
    (7, 14, 24, True, True), (2, 14, 5, False, True),            # other dt_rank classes
])
def test_fused_forward_equals_the_three_launches(B, rows, R, transposed, with_ln):
    from fastvim_amd import mixer_ops as M
    t = _inputs(B, rows, 384, R, seed=B * 31 + rows + R, with_ln=with_ln)
    ref = _three(t, rows, transposed, scaling=1.25)
    got = _one(t, rows, transposed, scaling=1.25)
    assert got is not None, "shape should take the fused launch"
    torch.cuda.synchronize()
    for n, a, b_ in zip(NAMES, got, ref):
        if b_ is None:
            assert a is None
            continue
        _close(n, a, b_)
    assert torch.equal(got[1], ref[1])           # skip: no product the compiler could contract differently
    again = _one(t, rows, transposed, scaling=1.25)
    for a, b_ in zip(got, again):
        assert b_ is None or torch.equal(a, b_)     # run to run: bit for bit
    assert M.mixer_mid_errors() == 0
    for f in M._MID_FLAGS.values():
        assert int(f.abs().sum()) == 0           # every flag consumed and taken down again


def test_fused_forward_replays_and_uneven_load():
    """200 launches back to back on changing inputs while a second stream keeps the chip busy with copies (uneven load,
    partner workgroups arriving at different times): every launch equals the three-launch result."""
    from fastvim_amd import mixer_ops as M
    ts = [_inputs(128, 14, 384, 12, seed=100 + k) for k in range(4)]
    refs = [_three(t, 14, bool(k & 1)) for k, t in enumerate(ts)]
    big = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)
    side = torch.cuda.Stream()
    for it in range(50):
        with torch.cuda.stream(side):
            big[: 1 << 27].copy_(big[1 << 27:])
        outs = [_one(t, 14, bool(k & 1)) for k, t in enumerate(ts)]
        if it == 0:
            first = outs
            for k in range(4):
                for n, a, b_ in zip(NAMES, outs[k], refs[k]):
                    _close(n, a, b_)
        for k in range(4):
            for n, a, b_ in zip(NAMES, outs[k], first[k]):
                assert torch.equal(a, b_), (it, k, n)          # a stale or torn hand-off would show as a changed bit
    torch.cuda.synchronize()
    assert M.mixer_mid_errors() == 0


def test_fused_forward_refuses_what_it_is_not_built_for():
    from fastvim_amd import _lib as L_
    lib = L_.lib()
    ok = lambda b, r, c, tpp, d, R, dt, pm: lib.fv_mixer_mid_fwd_ok(L_.i32(b), L_.i32(r), L_.i32(c), L_.i32(tpp), L_.i32(d), L_.i32(R), L_.i32(dt), L_.i32(pm))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert ok(cus // 2, 14, 14, 1, 384, 12, L_.FV_BF16, 0) == 1
    assert ok(cus // 2 + 1, 14, 14, 1, 384, 12, L_.FV_BF16, 0) == 0          # a pair must be resident together
    assert ok(8, 14, 14, 1, 384, 12, L_.FV_F32, 0) == 0 and ok(8, 14, 14, 1, 768, 24, L_.FV_BF16, 0) == 0
    assert ok(8, 14, 14, 8, 384, 12, L_.FV_BF16, 0) == 0 and ok(8, 14, 14, 1, 384, 12, L_.FV_BF16, 1) == 0
    assert ok(8, 128, 128, 1, 384, 12, L_.FV_BF16, 0) == 0


def test_mixer_module_takes_the_fused_launch_and_matches_the_unfused_build_of_itself(monkeypatch):
    """The FastVim-T mixer forward + backward through the fused launch against the same with the fused launch switched off
    (three launches): outputs and every gradient agree to bf16 rounding of the few differing activations."""
    from fastvim_amd import mixer_ops as M
    from fastvim_amd.mamba_simple_faster import Mamba
    torch.manual_seed(0)
    m = Mamba(192, token_size=(14, 14)).cuda()
    x = torch.randn(4, 196, 192, device="cuda")

    def run(fused):
        calls = []
        real = M.mixer_mid_fwd
        monkeypatch.setattr(M, "mixer_mid_fwd", (lambda *a, **k: (calls.append(1), real(*a, **k))[1]) if fused else (lambda *a, **k: None))
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.float().square().sum().backward()
        monkeypatch.setattr(M, "mixer_mid_fwd", real)
        return calls, y.detach(), xi.grad, [p.grad.clone() for p in m.parameters()]

    c1, y1, gx1, gp1 = run(True)
    c0, y0, gx0, gp0 = run(False)
    assert len(c1) == 1 and len(c0) == 0
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    assert rel(y1, y0) < 2e-3 and rel(gx1, gx0) < 2e-3
    for a, b_ in zip(gp1, gp0):
        assert rel(a, b_) < 5e-3
