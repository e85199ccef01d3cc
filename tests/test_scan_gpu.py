"""GPU parity of the HIP selective scan (through the C ABI) against the golden vectors
captured from the reference and against the CPU oracle.  Mirrors
mamba-1p1p1/tests/ops/test_selective_scan.py."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64
ACT = ("u", "delta", "B", "C", "z")


def _dev(d, dtype=None):
    out = {}
    for k, v in d.items():
        if v is None:
            out[k] = None
        else:
            v = v.cuda()
            if dtype is not None and k in ACT:
                v = v.to(dtype)
            out[k] = v
    return out


def _maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("case", sorted(load_golden("scan.pt").keys()))
def test_scan_fp32_vs_reference_golden(case):
    """BASELINE tolerance: forward within 1e-5 (relative to max|y|, see SURVEY 'Tolerance') in fp32."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    c = load_golden("scan.pt")[case]
    i = {k: (v.clone().requires_grad_() if v is not None else None) for k, v in _dev(c["inputs"]).items()}
    out, last = selective_scan_fn(i["u"], i["delta"], i["A"], i["B"], i["C"], i["D"], z=i["z"],
                                  delta_bias=i["delta_bias"], delta_softplus=c["softplus"],
                                  return_last_state=True)
    # fp64 oracle is the arbiter; the reference's own fp32 output is itself ~1e-7*max|y| away from it
    from oracle import selective_scan_oracle
    ci = c["inputs"]
    oref, lref = selective_scan_oracle(ci["u"], ci["delta"], ci["A"], ci["B"], ci["C"], ci["D"], ci["z"],
                                       ci["delta_bias"], c["softplus"], True, compute_dtype=F64, out_dtype=F64)
    scale = max(1.0, oref.abs().max().item())
    assert _maxerr(out, oref) <= 1e-5 * scale, (_maxerr(out, oref), scale)
    assert _maxerr(last, lref) <= 1e-5 * max(1.0, lref.abs().max().item())
    # and against the reference's own numbers with the reference's own tolerance
    assert torch.allclose(out.cpu(), c["out"], rtol=6e-4, atol=2e-3)
    assert torch.allclose(last.cpu(), c["last_state"], rtol=6e-4, atol=2e-3)
    out.backward(c["g"].cuda())
    rtol, atol, rtolw, atolw = 6e-4, 2e-3, 6e-4, 2e-3   # test_selective_scan.py:53-59 (has_z)
    tol = {"u": (rtol * 2, atol * 2), "delta": (rtol * 5, atol * 10), "A": (rtolw, atolw * 5),
           "B": (rtol, atol), "C": (rtol, atol), "D": (rtolw, atolw), "z": (rtolw, atolw),
           "delta_bias": (rtolw, atolw)}
    for k, gref in c["grads"].items():
        g = i[k].grad.cpu()
        rt, at = tol[k]
        assert torch.allclose(g, gref, rtol=rt, atol=at), (k, _maxerr(g, gref), gref.abs().max().item())
        # tighter: 2e-4 of the gradient's scale
        assert _maxerr(g, gref) <= 2e-4 * max(1.0, gref.abs().max().item()), (k, _maxerr(g, gref))


@pytest.mark.parametrize("case", ["b2_d4_L14_n8", "b2_d8_L14_n16", "b2_d4_L128_n8", "b2_d4_L256_n8", "b1_d2_L2100_n8"])
def test_scan_bf16_io(case):
    """bf16 I/O, fp32 math: rounded output within 1 bf16 ulp of the RNE-rounded fp64 oracle
    (BASELINE: 1e-3 on the pre-rounding value; 1 ulp = 2^-8 relative is the storage floor)."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    from oracle import selective_scan_oracle
    c = load_golden("scan.pt")[case]
    i = _dev(c["inputs"], torch.bfloat16)
    out = selective_scan_fn(i["u"], i["delta"], i["A"], i["B"], i["C"], i["D"], z=i["z"],
                            delta_bias=i["delta_bias"], delta_softplus=c["softplus"])
    assert out.dtype == torch.bfloat16
    ic = {k: (v.cpu() if v is not None else None) for k, v in i.items()}
    oref = selective_scan_oracle(ic["u"], ic["delta"], ic["A"], ic["B"], ic["C"], ic["D"], ic["z"],
                                 ic["delta_bias"], c["softplus"], compute_dtype=F64, out_dtype=F64)
    ulp = (oref.abs() * 2.0 ** -7).clamp_min(1e-30)
    # |rounded - exact| <= ulp/2 for an exact fp32 value; allow one extra ulp for fp32 accumulation
    bad = ((out.double().cpu() - oref).abs() > ulp).sum().item()
    assert bad == 0, bad
    # same inputs through the fp32 path: pre-rounding accumulator within 1e-3 relative
    o32 = selective_scan_fn(i["u"].float(), i["delta"].float(), i["A"], i["B"].float(), i["C"].float(), i["D"],
                            z=i["z"].float() if i["z"] is not None else None, delta_bias=i["delta_bias"],
                            delta_softplus=c["softplus"])
    assert _maxerr(o32, oref) <= 1e-3 * max(1.0, oref.abs().max().item())


@pytest.mark.parametrize("seqlen", [1, 3, 14, 64, 65, 196, 256, 257, 1024, 4096])
@pytest.mark.parametrize("varBC_groups", [1, 2])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
def test_scan_vs_oracle_seeded(seqlen, varBC_groups, itype):
    """The reference test's distributions (test_selective_scan.py:61-122) at more lengths,
    including ragged tile edges (1, 3, 65, 257) and multi-tile sequences; all three input types the reference's kernel
    dispatches (selective_scan.cpp:328-332: fp32 / fp16 / bf16, arithmetic always fp32)."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    from oracle import selective_scan_oracle
    torch.manual_seed(0)
    batch, dim, N = 2, 4, 8
    A = -0.5 * torch.rand(dim, N)
    shp = (batch, N, seqlen) if varBC_groups == 1 else (batch, varBC_groups, N, seqlen)
    B, C = torch.randn(*shp).to(itype), torch.randn(*shp).to(itype)
    D = torch.randn(dim)
    z = torch.randn(batch, dim, seqlen).to(itype)
    db = 0.5 * torch.rand(dim)
    u = torch.randn(batch, dim, seqlen).to(itype)
    delta = (0.5 * torch.rand(batch, dim, seqlen)).to(itype)
    cpu = dict(u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, delta_bias=db)
    leaves_c = {k: v.clone().requires_grad_() for k, v in cpu.items()}
    leaves_g = {k: v.clone().cuda().requires_grad_() for k, v in cpu.items()}
    oref, lref = selective_scan_oracle(*[leaves_c[k] for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")],
                                       True, True, compute_dtype=F64, out_dtype=F64)
    out, last = selective_scan_fn(*[leaves_g[k] for k in ("u", "delta", "A", "B", "C", "D")], z=leaves_g["z"],
                                  delta_bias=leaves_g["delta_bias"], delta_softplus=True, return_last_state=True)
    scale = max(1.0, oref.abs().max().item())
    ftol = {torch.float32: 2e-5, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[itype]      # one rounding of the output
    assert _maxerr(out, oref) <= ftol * scale
    assert _maxerr(last, lref) <= 2e-5 * max(1.0, lref.abs().max().item())
    g = torch.randn(batch, dim, seqlen, generator=torch.Generator().manual_seed(1))
    oref.backward(g.double())
    out.backward(g.cuda().to(itype))
    gtol = {torch.float32: 5e-4, torch.bfloat16: 3e-2, torch.float16: 4e-3}[itype]
    for k in cpu:
        gr, gg = leaves_c[k].grad, leaves_g[k].grad
        assert gg.dtype == cpu[k].dtype
        assert _maxerr(gg, gr) <= gtol * max(1.0, gr.abs().max().item()), (k, _maxerr(gg, gr), gr.abs().max().item())


def test_scan_constant_BC_and_options():
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    from oracle import selective_scan_oracle
    torch.manual_seed(3)
    batch, dim, N, L = 3, 6, 16, 50
    A = -0.5 * torch.rand(dim, N)
    for Bv, Cv, hasD, hasz, hasb, sp in [(False, False, True, False, True, True), (True, False, False, True, False, False),
                                         (False, True, True, True, True, True)]:
        B = torch.randn(batch, N, L) if Bv else torch.randn(dim, N)
        C = torch.randn(batch, N, L) if Cv else torch.randn(dim, N)
        cpu = dict(u=torch.randn(batch, dim, L), delta=0.5 * torch.rand(batch, dim, L), A=A, B=B, C=C,
                   D=torch.randn(dim) if hasD else None, z=torch.randn(batch, dim, L) if hasz else None,
                   delta_bias=0.5 * torch.rand(dim) if hasb else None)
        lc = {k: (v.clone().requires_grad_() if v is not None else None) for k, v in cpu.items()}
        lg = {k: (v.clone().cuda().requires_grad_() if v is not None else None) for k, v in cpu.items()}
        o = selective_scan_oracle(lc["u"], lc["delta"], lc["A"], lc["B"], lc["C"], lc["D"], lc["z"], lc["delta_bias"],
                                  sp, compute_dtype=F64, out_dtype=F64)
        y = selective_scan_fn(lg["u"], lg["delta"], lg["A"], lg["B"], lg["C"], lg["D"], z=lg["z"],
                              delta_bias=lg["delta_bias"], delta_softplus=sp)
        assert _maxerr(y, o) <= 2e-5 * max(1.0, o.abs().max().item())
        g = torch.randn(batch, dim, L)
        o.backward(g.double()); y.backward(g.cuda())
        for k in cpu:
            if cpu[k] is not None:
                assert _maxerr(lg[k].grad, lc[k].grad) <= 5e-4 * max(1.0, lc[k].grad.abs().max().item()), k


def test_scan_full_size_properties():
    """BASELINE config shapes (B, d_in, Lc, N): no oracle at this size on the hot path of the test,
    instead size-independent properties: linearity in u, and chunk-consistency (a scan of
    [first half | second half] equals scan(second half) started from last_state(first half))."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    torch.manual_seed(0)
    for (Bsz, D, L, N) in [(128, 384, 14, 16), (8, 1536, 128, 16), (64, 768, 112, 16)]:
        dev = "cuda"
        u1, u2 = torch.randn(Bsz, D, L, device=dev), torch.randn(Bsz, D, L, device=dev)
        delta = 0.5 * torch.rand(Bsz, D, L, device=dev)
        A = -0.5 * torch.rand(D, N, device=dev)
        Bm, Cm = torch.randn(Bsz, N, L, device=dev), torch.randn(Bsz, N, L, device=dev)
        db = 0.5 * torch.rand(D, device=dev)
        f = lambda u: selective_scan_fn(u, delta, A, Bm, Cm, None, None, db, True)
        y1, y2, y12 = f(u1), f(u2), f(2.0 * u1 - 3.0 * u2)
        lin = (y12 - (2.0 * y1 - 3.0 * y2)).abs().max().item()
        assert lin <= 1e-4 * max(1.0, y12.abs().max().item()), lin
        # deterministic: bitwise identical on a re-run, forward and backward
        u = u1.clone().requires_grad_()
        A_ = A.clone().requires_grad_()
        Bm_ = Bm.clone().requires_grad_()
        ya = selective_scan_fn(u, delta, A_, Bm_, Cm, None, None, db, True)
        ga = torch.autograd.grad(ya, (u, A_, Bm_), torch.ones_like(ya))
        yb = selective_scan_fn(u, delta, A_, Bm_, Cm, None, None, db, True)
        gb = torch.autograd.grad(yb, (u, A_, Bm_), torch.ones_like(yb))
        assert torch.equal(ya, yb) and all(torch.equal(a, b) for a, b in zip(ga, gb))


def test_scan_errors():
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    u = torch.randn(2, 4, 8, device="cuda")
    A = -torch.rand(4, 300, device="cuda")
    with pytest.raises(RuntimeError):
        selective_scan_fn(u, u, A, torch.randn(2, 300, 8, device="cuda"), torch.randn(2, 300, 8, device="cuda"))
    with pytest.raises(RuntimeError):
        selective_scan_fn(u.cpu(), u.cpu(), A[:, :8].cpu(), torch.randn(2, 8, 8), torch.randn(2, 8, 8))


@pytest.mark.parametrize("Bsz,D,L", [(2, 64, 33), (2, 128, 37), (1, 64, 128), (3, 64, 112), (2, 96, 100), (2, 64, 64),
                                     (3, 3072, 40), (130, 64, 35)])
@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
def test_short_scan_segmented_forward_vs_oracle(Bsz, D, L, itype):
    """32 < L <= 128 with d_state 16 on FEW rows (at most 128 blocks of 64 channels) takes the time-segmented forward
    kernel (scan_short_fwd_seg_kernel: a block's four waves scan four consecutive segments -- end states from zero, a fold
    over the earlier segments, a second scan from the true entry state; ragged last segment when L % 4 != 0); the last two
    shapes (144 / 130 blocks) take the serial quad-sharing kernel at those lengths (round 6: the threshold moved): output,
    gate, skip, bias, softplus and last state against the fp64 oracle, and the gradients through the chunked backward
    kernel (three and two chunks: checkpoints, ragged last chunk)."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    from oracle import selective_scan_oracle
    g = torch.Generator().manual_seed(Bsz * 1000 + D + L)
    N = 16
    cpu = dict(u=torch.randn(Bsz, D, L, generator=g).to(itype), delta=(0.5 * torch.rand(Bsz, D, L, generator=g)).to(itype),
               A=-0.5 * torch.rand(D, N, generator=g), B=torch.randn(Bsz, N, L, generator=g).to(itype),
               C=torch.randn(Bsz, N, L, generator=g).to(itype), D=torch.randn(D, generator=g),
               z=torch.randn(Bsz, D, L, generator=g).to(itype), delta_bias=0.5 * torch.rand(D, generator=g))
    lc = {k: v.clone().requires_grad_() for k, v in cpu.items()}
    lg = {k: v.clone().cuda().requires_grad_() for k, v in cpu.items()}
    order = ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")
    oref, lref = selective_scan_oracle(*[lc[k] for k in order], True, True, compute_dtype=F64, out_dtype=F64)
    out, last = selective_scan_fn(*[lg[k] for k in order[:6]], z=lg["z"], delta_bias=lg["delta_bias"], delta_softplus=True,
                                  return_last_state=True)
    assert out.dtype == itype
    ftol = {torch.float32: 2e-5, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[itype]
    assert _maxerr(out, oref) <= ftol * max(1.0, oref.abs().max().item())
    assert _maxerr(last, lref) <= 2e-5 * max(1.0, lref.abs().max().item())
    out2, _ = selective_scan_fn(*[lg[k] for k in order[:6]], z=lg["z"], delta_bias=lg["delta_bias"], delta_softplus=True,
                                return_last_state=True)
    assert torch.equal(out, out2)
    go = torch.randn(Bsz, D, L, generator=torch.Generator().manual_seed(1))
    oref.backward(go.double())
    out.backward(go.cuda().to(itype))
    gtol = {torch.float32: 5e-4, torch.bfloat16: 3e-2, torch.float16: 4e-3}[itype]
    for k in cpu:
        gr, gg = lc[k].grad, lg[k].grad
        assert _maxerr(gg, gr) <= gtol * max(1.0, gr.abs().max().item()), (k, _maxerr(gg, gr), gr.abs().max().item())


@pytest.mark.parametrize("itype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seqlen", [14, 200])
def test_scan_denormal_range_inputs_vs_oracle(itype, seqlen):
    """The scan / conv / mixer kernels are built with -fgpu-flush-denormals-to-zero (fastvim_amd/build.py; the reference's
    extension is built with nvcc --use_fast_math, which flushes as well: mamba-1p1p1/setup.py:131-149).  Feed the kernel
    inputs whose intermediate products live in the fp32 denormal range -- tiny steps delta in [1e-6, 1e-3], A in [-16, -1],
    and u a mixture of O(1) values and values of 1e-20 ... 1e-30, so that delta * B * u and C * x are denormal for part of
    the channels -- and hold it to the same bound as everywhere else: 1e-5 * max(1, max|y|) against the fp64 oracle (fp32
    I/O; one storage ulp for the 16-bit types).  A flushed denormal is an absolute error below 1.2e-38 per term, so the bound
    holds iff nothing ELSE depends on denormal arithmetic (e.g. a scaled exp / log expansion)."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    from oracle import selective_scan_oracle
    g = torch.Generator().manual_seed(seqlen)
    Bsz, D, N = 2, 64, 16
    u = torch.randn(Bsz, D, seqlen, generator=g)
    tiny = 10.0 ** (-20.0 - 10.0 * torch.rand(Bsz, D, 1, generator=g))
    u = torch.where(torch.rand(Bsz, D, 1, generator=g) < 0.5, u * tiny, u)            # half of the channels: 1e-20 .. 1e-30
    delta = 10.0 ** (-6.0 + 3.0 * torch.rand(Bsz, D, seqlen, generator=g))            # [1e-6, 1e-3]
    A = -(1.0 + 15.0 * torch.rand(D, N, generator=g))                                 # [-16, -1]
    Bm = torch.randn(Bsz, 1, N, seqlen, generator=g)
    Cm = torch.randn(Bsz, 1, N, seqlen, generator=g)
    Dv = torch.randn(D, generator=g)
    z = torch.randn(Bsz, D, seqlen, generator=g)
    if itype != torch.float32:       # the inputs as the 16-bit kernel sees them (fp16 flushes 1e-20 itself: keep those exact zeros)
        u, delta, Bm, Cm, z = (t.to(itype).float() for t in (u, delta, Bm, Cm, z))
    dev = lambda t, act=True: t.cuda().to(itype) if act else t.cuda()
    out, last = selective_scan_fn(dev(u), dev(delta), dev(A, False), dev(Bm), dev(Cm), dev(Dv, False), z=dev(z),
                                  delta_bias=None, delta_softplus=False, return_last_state=True)
    oref, lref = selective_scan_oracle(u, delta, A, Bm, Cm, Dv, z, None, False, True, compute_dtype=F64, out_dtype=F64)
    scale = max(1.0, oref.abs().max().item())
    if itype == torch.float32:
        assert _maxerr(out, oref) <= 1e-5 * scale, (_maxerr(out, oref), scale)
        # the tiny channels on their own scale as well: relative error of what is NOT denormal stays small
        big = oref.abs() > 1e-30
        rel = ((out.double().cpu() - oref).abs() / oref.abs().clamp_min(1e-300))[big]
        assert rel.max().item() <= 1e-3, rel.max().item()
    else:
        ulp = 2.0 ** (-8 if itype == torch.bfloat16 else -11)
        assert ((out.double().cpu() - oref).abs() <= ulp * oref.abs() + 1e-5 * scale).all()
    assert _maxerr(last, lref) <= 1e-5 * max(1.0, lref.abs().max().item())
    assert torch.isfinite(out.float()).all()
