"""GPU parity of the remaining reference op surfaces: causal_conv1d_fn
(PyPI causal-conv1d 1.1.3.post1, pinned to the F.conv1d formula the reference itself falls back to),
the compressed-scan fork op, and the fused FastVim_mamba_inner_fn_no_out_proj_withoutZ."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("case", sorted(load_golden("conv1d.pt").keys()))
def test_causal_conv1d_fp32_vs_reference_golden(case):
    from fastvim_amd.causal_conv1d import causal_conv1d_fn
    c = load_golden("conv1d.pt")[case]
    x = c["x"].cuda().requires_grad_()
    w = c["w"].cuda().requires_grad_()
    b = c["b"].cuda().requires_grad_() if c["b"] is not None else None
    y = causal_conv1d_fn(x, w, b, activation=c["act"])
    assert _err(y, c["y"]) <= 2e-6 * max(1.0, c["y"].abs().max().item())
    y.backward(c["g"].cuda())
    assert _err(x.grad, c["dx"]) <= 5e-6 * max(1.0, c["dx"].abs().max().item())
    assert _err(w.grad, c["dw"]) <= 2e-5 * max(1.0, c["dw"].abs().max().item())
    if b is not None:
        assert _err(b.grad, c["db"]) <= 2e-5 * max(1.0, c["db"].abs().max().item())


@pytest.mark.parametrize("shape,width,act,dtype", [
    ((128, 384, 196), 4, "silu", torch.bfloat16),       # FastVim-T mixer shape (config 2)
    ((2, 1536, 1000), 4, "silu", torch.float32),
    ((3, 7, 1), 2, None, torch.float32),
    ((2, 16, 1025), 3, "swish", torch.float16),
    ((1, 5, 4099), 4, None, torch.bfloat16),
])
def test_causal_conv1d_vs_oracle(shape, width, act, dtype):
    from fastvim_amd.causal_conv1d import causal_conv1d_fn
    from oracle import causal_conv1d_oracle
    g = torch.Generator().manual_seed(shape[2] + width)
    x = torch.randn(shape, generator=g).to(dtype)
    w = 0.5 * torch.randn(shape[1], width, generator=g)
    b = 0.1 * torch.randn(shape[1], generator=g)
    go = torch.randn(shape, generator=g).to(dtype)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = causal_conv1d_oracle(xr, wr, br, act, compute_dtype=F64, out_dtype=F64)
    yr.backward(go.double())
    xg, wg, bg = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    y = causal_conv1d_fn(xg, wg, bg, activation=act)
    assert y.dtype == dtype and y.shape == x.shape
    lo = dtype != torch.float32
    ty = 1e-2 if lo else 2e-6
    assert _err(y, yr) <= ty * max(1.0, yr.abs().max().item()), _err(y, yr)
    y.backward(go.cuda())
    assert _err(xg.grad, xr.grad) <= ty * max(1.0, xr.grad.abs().max().item())
    tw = 2e-5 * (shape[0] * shape[2]) ** 0.5 if not lo else 1e-2
    assert _err(wg.grad, wr.grad) <= tw * max(1.0, wr.grad.abs().max().item()), _err(wg.grad, wr.grad)
    assert _err(bg.grad, br.grad) <= tw * max(1.0, br.grad.abs().max().item())
    # deterministic gradients (fixed-order reductions)
    xg2, wg2 = x.cuda().requires_grad_(), w.cuda().requires_grad_()
    causal_conv1d_fn(xg2, wg2, bg.detach(), activation=act).backward(go.cuda())
    assert torch.equal(wg2.grad, wg.grad) and torch.equal(xg2.grad, xg.grad)


def test_causal_conv1d_errors():
    from fastvim_amd.causal_conv1d import causal_conv1d_fn
    x = torch.randn(2, 4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="width between 2 and 4"):
        causal_conv1d_fn(x, torch.randn(4, 5, device="cuda"))
    with pytest.raises(NotImplementedError):
        causal_conv1d_fn(x, torch.randn(4, 4, device="cuda"), activation="relu")
    with pytest.raises(RuntimeError):
        causal_conv1d_fn(x, torch.randn(3, 4, device="cuda"))


@pytest.mark.parametrize("case", sorted(load_golden("compressed_scan.pt").keys()))
def test_compressed_scan_vs_reference_golden(case):
    """Golden vectors captured from the fork's own `selective_scan_ref`
    (fastvim_kernel/.../faster_mamba_ssm/ops/selective_scan_interface.py:188-252)."""
    from fastvim_amd.selective_scan_interface import compressed_selective_scan_fn
    c = load_golden("compressed_scan.pt")[case]
    i = {k: (v.cuda() if v is not None else None) for k, v in c["inputs"].items()}
    out, last = compressed_selective_scan_fn(i["u_full"], i["u_c"], i["delta"], i["A"], i["B"], i["C"], i["D"],
                                             None, i["delta_bias"], False, True)
    assert _err(out, c["out"]) <= 1e-5 * max(1.0, c["out"].abs().max().item())
    assert _err(last, c["last_state"]) <= 1e-5 * max(1.0, c["last_state"].abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_compressed_scan_backward_vs_oracle(dtype):
    """The compressed scan with the backward the fork lacks, for every storage type of the op-level API (the fork itself
    is fp32-only, fastvim_kernel/.../selective_scan.cpp:345-348; fp16 / bf16 follow selective_scan_fn's rules: inputs and
    output in the storage type, arithmetic and parameter gradients fp32)."""
    from fastvim_amd.selective_scan_interface import compressed_selective_scan_fn
    from oracle.scan import compressed_scan_oracle
    g = torch.Generator().manual_seed(3)
    Bsz, D, Lc, cf, N = 2, 8, 14, 14, 16
    lo = lambda v: v.to(dtype).float()
    t = dict(u=lo(torch.randn(Bsz, D, Lc * cf, generator=g)), u_c=lo(torch.randn(Bsz, D, Lc, generator=g)),
             delta=lo(0.5 * torch.rand(Bsz, D, Lc, generator=g)), A=-0.5 * torch.rand(D, N, generator=g),
             B=lo(torch.randn(Bsz, N, Lc, generator=g)), C=lo(torch.randn(Bsz, N, Lc, generator=g)),
             D=torch.randn(D, generator=g), bias=0.5 * torch.rand(D, generator=g))
    go = lo(torch.randn(Bsz, D, Lc * cf, generator=g))
    r = {k: v.double().requires_grad_() for k, v in t.items()}
    yr = compressed_scan_oracle(r["u"], r["u_c"], r["delta"], r["A"], r["B"], r["C"], r["D"], r["bias"], True,
                                compute_dtype=F64, out_dtype=F64)
    yr.backward(go.double())
    stored = ("u", "u_c", "delta", "B", "C")
    q = {k: (v.to(dtype) if k in stored else v).cuda().requires_grad_() for k, v in t.items()}
    y = compressed_selective_scan_fn(q["u"], q["u_c"], q["delta"], q["A"], q["B"], q["C"], q["D"], None, q["bias"], True)
    assert y.dtype == dtype
    ty, tg = {torch.float32: (2e-5, 1e-4), torch.float16: (2.0 ** -10, 4e-3), torch.bfloat16: (2.0 ** -7, 3e-2)}[dtype]
    assert _err(y, yr) <= ty * max(1.0, yr.abs().max().item())
    y.backward(go.to(dtype).cuda())
    for k in t:
        assert q[k].grad.dtype == q[k].dtype
        e = _err(q[k].grad, r[k].grad)
        assert e <= tg * max(1.0, r[k].grad.abs().max().item()), (k, e)


def test_fp16_autocast_and_half_models_run_through_the_module_path():
    """The reference's other mixed precision (``--precision 16-mixed``, imagenet_classification/train.py:17) and a
    ``model.half()`` (its kernels dispatch fp16: selective_scan.cpp:328-332): the module path computes them in fp32 with
    fp16 at the model boundary (``mamba_simple_faster._compute_dtype``) -- same function as the fp32 run up to the fp16
    rounding of inputs / parameters / outputs, gradients flow (GradScaler-style scaled loss included), no library GEMM is
    involved (the fp32-MFMA kernels run)."""
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.mamba_simple_faster import Mamba
    torch.manual_seed(0)
    m = VisionMamba(img_size=64, patch_size=16, depth=2, embed_dim=64, num_classes=10, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, final_pool_type="mean", drop_path_rate=0.0).cuda()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    ref = m(x)                                                  # fp32 run
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(x)
    assert y.dtype == torch.float16 and torch.isfinite(y.float()).all()
    assert (y.float() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    # training under fp16 autocast with a scaled loss: finite gradients equal to the fp32 run's (x the scale)
    m.zero_grad()
    ref.float().square().mean().backward()
    g32 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        loss = m(x).float().square().mean()
    (loss * 1024.0).backward()
    for n, p in m.named_parameters():
        assert torch.isfinite(p.grad).all(), n
        s_ = max(g32[n].abs().max().item(), 1e-6)
        assert (p.grad / 1024.0 - g32[n]).abs().max().item() <= 2e-2 * s_, n
    # a .half() model on fp16 images (inference): fp16 out, close to the fp32 model evaluated on the same rounded inputs
    import copy
    mh = copy.deepcopy(m).half().eval()
    mr = copy.deepcopy(mh).float().eval()                        # fp16-rounded parameters, fp32 arithmetic
    with torch.no_grad():
        yh = mh(x.half())
        yr = mr(x.half().float())
    assert yh.dtype == torch.float16
    assert (yh.float() - yr).abs().max().item() <= 2e-3 * max(1.0, yr.abs().max().item())
    # the mixer on its own: fp16 activations in -> fp16 out; fp16 parameters get fp16 gradients
    mx = Mamba(64, token_size=(4, 4)).cuda()
    h = torch.randn(2, 16, 64, device="cuda")
    y32 = mx(h)
    mxh = copy.deepcopy(mx).half()
    hh = h.half().requires_grad_()
    yh = mxh(hh)
    assert yh.dtype == torch.float16 and (yh.float() - y32).abs().max().item() <= 1e-2 * max(1.0, y32.abs().max().item())
    yh.float().sum().backward()
    assert hh.grad.dtype == torch.float16 and all(p.grad is not None and p.grad.dtype == torch.float16 for p in mxh.parameters())
    with torch.autocast("cuda", dtype=torch.float16):
        ya = mx(h)
    assert (ya.float() - y32).abs().max().item() <= 1e-5 * max(1.0, y32.abs().max().item())      # fp32 inside
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.isfinite(m(x).float()).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_inner_fn_vs_mixer_oracle_direction(dtype):
    """FastVim_mamba_inner_fn_no_out_proj_withoutZ == one direction of the mixer: compare with the fp64
    oracle's forward-direction branch (conv -> pool -> projections -> scan -> expand -> + D x)."""
    from fastvim_amd.selective_scan_interface import FastVim_mamba_inner_fn_no_out_proj_withoutZ as fused
    from oracle import causal_conv1d_oracle, selective_scan_oracle
    g = torch.Generator().manual_seed(11)
    Bsz, d_in, rows, cols, N, R = 2, 64, 6, 4, 16, 4
    x = torch.randn(Bsz, d_in, rows * cols, generator=g).to(dtype).float()
    cw = 0.5 * torch.randn(d_in, 1, 4, generator=g)
    cb = 0.1 * torch.randn(d_in, generator=g)
    Wx = torch.randn(R + 2 * N, d_in, generator=g) * d_in ** -0.5
    Wdt = torch.randn(d_in, R, generator=g) * R ** -0.5
    A = -torch.exp(torch.log(torch.arange(1, N + 1).float())[None].repeat(d_in, 1))
    Dp = 1 + 0.1 * torch.randn(d_in, generator=g)
    bias = torch.rand(d_in, generator=g) * 0.1
    go = torch.randn(Bsz, d_in, rows * cols, generator=g).to(dtype).float()
    leaves = dict(x=x, cw=cw, cb=cb, Wx=Wx, Wdt=Wdt, A=A, D=Dp, bias=bias)
    r = {k: v.double().requires_grad_() for k, v in leaves.items()}
    conv = causal_conv1d_oracle(r["x"], r["cw"].reshape(d_in, 4), r["cb"], "silu", compute_dtype=F64, out_dtype=F64)
    pooled = conv.reshape(Bsz, d_in, rows, cols).mean(3)
    x_dbl = pooled.transpose(1, 2).reshape(Bsz * rows, d_in) @ r["Wx"].t()
    delta = (r["Wdt"] @ x_dbl[:, :R].t()).view(d_in, Bsz, rows).transpose(0, 1)
    Bm = x_dbl[:, R:R + N].view(Bsz, rows, N).transpose(1, 2)
    Cm = x_dbl[:, -N:].view(Bsz, rows, N).transpose(1, 2)
    yc = selective_scan_oracle(pooled, delta, r["A"], Bm, Cm, None, None, r["bias"], True, False, F64, F64)
    yr = yc.repeat_interleave(cols, 2) + r["D"][None, :, None] * conv
    yr.backward(go.double())
    q = {k: v.cuda().requires_grad_() for k, v in leaves.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        y = fused(q["x"].to(dtype) if dtype != torch.float32 else q["x"], q["cw"], q["cb"], q["Wx"], q["Wdt"], q["A"],
                  None, None, q["D"], q["bias"], None, None, True, cols, "mean", 1, (Bsz, d_in, rows, cols))
    lo = dtype != torch.float32
    assert _err(y, yr) <= (3e-2 if lo else 2e-5) * max(1.0, yr.abs().max().item()), _err(y, yr)
    y.backward(go.cuda().to(y.dtype))
    for k in leaves:
        e = _err(q[k].grad, r[k].grad)
        assert e <= (5e-2 if lo else 2e-4) * max(1.0, r[k].grad.abs().max().item()), (k, e, r[k].grad.abs().max().item())


@pytest.mark.parametrize("dtype,L", [(torch.float32, 197), (torch.bfloat16, 197), (torch.float32, 24)])
def test_vim_inner_fn_vs_oracle(dtype, L):
    """mamba_inner_fn_no_out_proj_withoutZ (the Vim baseline's fused op, selective_scan_interface.py:779-1016,
    1684-1713): conv + SiLU -> x_proj / dt_proj -> scan over all L tokens with D inside, against the fp64 oracle
    (197 = 196 patches + the class token of Vim-T at 224 px)."""
    from fastvim_amd.selective_scan_interface import mamba_inner_fn_no_out_proj_withoutZ as fused
    from oracle import causal_conv1d_oracle, selective_scan_oracle
    g = torch.Generator().manual_seed(13)
    Bsz, d_in, N, R = 2, 64, 16, 4
    x = torch.randn(Bsz, d_in, L, generator=g).to(dtype).float()
    cw = 0.5 * torch.randn(d_in, 1, 4, generator=g)
    cb = 0.1 * torch.randn(d_in, generator=g)
    Wx = torch.randn(R + 2 * N, d_in, generator=g) * d_in ** -0.5
    Wdt = torch.randn(d_in, R, generator=g) * R ** -0.5
    A = -torch.exp(torch.log(torch.arange(1, N + 1).float())[None].repeat(d_in, 1))
    Dp = 1 + 0.1 * torch.randn(d_in, generator=g)
    bias = torch.rand(d_in, generator=g) * 0.1 - 3.0
    go = torch.randn(Bsz, d_in, L, generator=g).to(dtype).float()
    leaves = dict(x=x, cw=cw, cb=cb, Wx=Wx, Wdt=Wdt, A=A, D=Dp, bias=bias)
    r = {k: v.double().requires_grad_() for k, v in leaves.items()}
    conv = causal_conv1d_oracle(r["x"], r["cw"].reshape(d_in, 4), r["cb"], "silu", compute_dtype=F64, out_dtype=F64)
    x_dbl = conv.transpose(1, 2).reshape(Bsz * L, d_in) @ r["Wx"].t()
    delta = (r["Wdt"] @ x_dbl[:, :R].t()).view(d_in, Bsz, L).transpose(0, 1)
    Bm = x_dbl[:, R:R + N].view(Bsz, L, N).transpose(1, 2)
    Cm = x_dbl[:, -N:].view(Bsz, L, N).transpose(1, 2)
    yr = selective_scan_oracle(conv, delta, r["A"], Bm, Cm, r["D"], None, r["bias"], True, False, F64, F64)
    yr.backward(go.double())
    q = {k: v.cuda().requires_grad_() for k, v in leaves.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        y = fused(q["x"].to(dtype) if dtype != torch.float32 else q["x"], q["cw"], q["cb"], q["Wx"], q["Wdt"], q["A"],
                  None, None, q["D"], q["bias"], None, None, True)
    lo = dtype != torch.float32
    assert y.shape == (Bsz, d_in, L)
    assert _err(y, yr) <= (3e-2 if lo else 2e-5) * max(1.0, yr.abs().max().item()), _err(y, yr)
    y.backward(go.cuda().to(y.dtype))
    for k in leaves:
        e = _err(q[k].grad, r[k].grad)
        assert e <= (5e-2 if lo else 2e-4) * max(1.0, r[k].grad.abs().max().item()), (k, e, r[k].grad.abs().max().item())


def test_fused_inner_fn_is_one_node_with_proj_biases_scaling_and_given_C():
    """The reference's fused op is ONE autograd node that keeps x and x_dbl and re-derives conv_out / delta in backward
    (selective_scan_interface.py:452-776, checkpoint_lvl 1); same here.  Arguments the other tests leave at their
    defaults: B_proj_bias, scaling_factor != 1 (:505-506), a constant C handed in (:549-551) -- against fp64 autograd
    of the same formula."""
    from fastvim_amd.selective_scan_interface import FastVim_mamba_inner_fn_no_out_proj_withoutZ as fused
    from oracle import causal_conv1d_oracle, selective_scan_oracle
    g = torch.Generator().manual_seed(17)
    Bsz, d_in, rows, cols, N, R = 3, 32, 5, 7, 16, 2
    x = torch.randn(Bsz, d_in, rows * cols, generator=g)
    leaves = dict(x=x, cw=0.5 * torch.randn(d_in, 1, 4, generator=g), cb=0.1 * torch.randn(d_in, generator=g),
                  Wx=torch.randn(R + 2 * N, d_in, generator=g) * d_in ** -0.5, Wdt=torch.randn(d_in, R, generator=g) * R ** -0.5,
                  A=-torch.exp(torch.log(torch.arange(1, N + 1).float())[None].repeat(d_in, 1)),
                  C=torch.randn(d_in, N, generator=g), D=1 + 0.1 * torch.randn(d_in, generator=g),
                  bias=torch.rand(d_in, generator=g) * 0.1, Bpb=0.2 * torch.randn(N, generator=g))
    go = torch.randn(Bsz, d_in, rows * cols, generator=g)
    sf = 0.5
    r = {k: v.double().requires_grad_() for k, v in leaves.items()}
    conv = causal_conv1d_oracle(r["x"], r["cw"].reshape(d_in, 4), r["cb"], "silu", compute_dtype=F64, out_dtype=F64)
    pooled = conv.reshape(Bsz, d_in, rows, cols).mean(3) * sf
    x_dbl = pooled.transpose(1, 2).reshape(Bsz * rows, d_in) @ r["Wx"].t()
    delta = (r["Wdt"] @ x_dbl[:, :R].t()).view(d_in, Bsz, rows).transpose(0, 1)
    Bm = (x_dbl[:, R:R + N] + r["Bpb"]).view(Bsz, rows, N).transpose(1, 2)
    yc = selective_scan_oracle(pooled, delta, r["A"], Bm, r["C"], None, None, r["bias"], True, False, F64, F64)
    yr = yc.repeat_interleave(cols, 2) + r["D"][None, :, None] * conv
    yr.backward(go.double())
    q = {k: v.cuda().requires_grad_() for k, v in leaves.items()}
    y = fused(q["x"], q["cw"], q["cb"], q["Wx"], q["Wdt"], q["A"], None, q["C"], q["D"], q["bias"], q["Bpb"], None, True,
              cols, "mean", sf, (Bsz, d_in, rows, cols))
    assert type(y.grad_fn).__name__ == "_InnerFnNoOutProjWithoutZBackward"
    kept = [t for t in y.grad_fn.saved_tensors if t.numel()]
    assert not any(t.shape == y.shape and t.data_ptr() != q["x"].data_ptr() for t in kept), "conv_out must be re-derived"
    assert _err(y, yr) <= 2e-5 * max(1.0, yr.abs().max().item()), _err(y, yr)
    y.backward(go.cuda())
    for k in leaves:
        e = _err(q[k].grad, r[k].grad)
        assert e <= 2e-4 * max(1.0, r[k].grad.abs().max().item()), (k, e, r[k].grad.abs().max().item())
    # bitwise deterministic (fixed-order split-K and batch sums)
    q2 = {k: v.cuda().requires_grad_() for k, v in leaves.items()}
    y2 = fused(q2["x"], q2["cw"], q2["cb"], q2["Wx"], q2["Wdt"], q2["A"], None, q2["C"], q2["D"], q2["bias"], q2["Bpb"], None,
               True, cols, "mean", sf, (Bsz, d_in, rows, cols))
    y2.backward(go.cuda())
    assert torch.equal(y, y2) and all(torch.equal(q[k].grad, q2[k].grad) for k in leaves)


@pytest.mark.parametrize("Mrows,d_in,W", [(1792, 384, 44), (37, 768, 56), (256, 1536, 80), (16, 64, 34), (100, 2560, 112)])
def test_xproj_fwd_kernel_vs_torch(Mrows, d_in, W):
    """fv_mixer_xproj_fwd (both directions in one launch) against an fp64 product of the same bf16 operands."""
    from fastvim_amd import mixer_ops as M
    torch.manual_seed(W)
    xc = torch.randn(2, 1, Mrows, d_in, device="cuda").bfloat16()
    Wx = (torch.randn(2, W, d_in, device="cuda") * d_in ** -0.5).bfloat16()
    from fastvim_amd import _lib as L
    out = torch.empty(2, Mrows, W, device="cuda", dtype=torch.bfloat16)      # the C entry point itself (the Python
    rc = L.lib().fv_mixer_xproj_fwd(L.ptr(xc), L.ptr(Wx), L.ptr(out), L.i32(Mrows), L.i32(d_in), L.i32(W),   # wrapper
                                    L.stream_of(xc))                          # routes wide models to hipBLASLt)
    L.check(rc, "mixer_xproj_fwd")
    assert torch.equal(M.xproj_fwd(xc, Wx), out) or d_in > 512
    ref = torch.bmm(xc.view(2, Mrows, d_in).double(), Wx.double().transpose(1, 2))
    assert out.shape == (2, Mrows, W) and out.dtype == torch.bfloat16
    assert (out.double() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())


def test_deferred_partial_reductions_match_single_launches():
    """fv_reduce_partials_multi (up to 208 queued gradient-partial reductions in one launch) against fp64 and against
    the one-job kernel, which shares its summation code (bitwise equal: an eager model and one on the flat training
    state stay in lock-step); a repeat is bitwise identical."""
    from fastvim_amd import mixer_ops as M
    torch.manual_seed(0)
    shapes = [(7, 768 * 192), (256, 768), (256, 4608), (37, 384 * 29), (1, 1000), (9, 44 * 384), (64, 20)]
    for ragged in (False, True):          # odd element counts: no alignment assumptions
        parts = [torch.randn(S, n + (1 if ragged and k == 2 else 0), device="cuda") for k, (S, n) in enumerate(shapes)]
        base = [torch.randn(p.shape[1], device="cuda") for p in parts]
        single = [b.clone() for b in base]
        for p, o in zip(parts, single):
            M.reduce_partials(p, p.shape[0], out=o, accumulate=True, defer=False)
        multi = [b.clone() for b in base]
        M.defer_reductions(True)
        try:
            for p, o in zip(parts, multi):
                M.reduce_partials(p, p.shape[0], out=o, accumulate=True)
            M.flush_reductions()
        finally:
            M.defer_reductions(False)
        for p, b, s_, m_ in zip(parts, base, single, multi):
            ref = b.double() + p.double().sum(0)
            tol = 1e-5 * max(1.0, ref.abs().max().item())
            assert (m_.double() - ref).abs().max().item() <= tol and torch.equal(s_, m_)
        again = [b.clone() for b in base]
        M.defer_reductions(True)
        try:
            for p, o in zip(parts, again):
                M.reduce_partials(p, p.shape[0], out=o, accumulate=True)
            M.flush_reductions()
        finally:
            M.defer_reductions(False)
        assert all(torch.equal(a_, m_) for a_, m_ in zip(again, multi))


@pytest.mark.parametrize("B,C,dtype", [(128, 1000, torch.bfloat16), (128, 1000, torch.float32), (5, 10, torch.float32),
                                       (3, 2048, torch.bfloat16)])
def test_soft_target_cross_entropy_vs_oracle(B, C, dtype):
    """fastvim_amd.losses.SoftTargetCrossEntropy (fv_soft_target_ce: value + gradient in one launch) against the fp64
    restatement of timm's formula, mixup-style soft targets (rows sum to 1) and un-normalised ones; upstream gradient
    scale honoured; repeat bitwise identical."""
    from fastvim_amd.losses import SoftTargetCrossEntropy
    from oracle import soft_target_ce_oracle
    torch.manual_seed(B + C)
    x = (torch.randn(B, C, device="cuda") * 3).to(dtype).requires_grad_(True)
    for normalised in (True, False):
        t = torch.rand(B, C, device="cuda")
        if normalised:
            t = t / t.sum(-1, keepdim=True)
        ref, gref = soft_target_ce_oracle(x, t)
        x.grad = None
        loss = SoftTargetCrossEntropy()(x, t)
        (loss * 2.5).backward()
        assert loss.dtype == torch.float32 and loss.dim() == 0
        assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
        tol = (1e-6 if dtype == torch.float32 else 2e-2) * gref.abs().max().item() * 2.5
        assert (x.grad.double().cpu() - 2.5 * gref).abs().max().item() <= tol
        g1 = x.grad.clone()
        x.grad = None
        loss2 = SoftTargetCrossEntropy()(x, t)
        (loss2 * 2.5).backward()
        assert torch.equal(loss, loss2) and torch.equal(g1, x.grad)


def test_deferred_reductions_across_the_job_table_boundary():
    """260 queued reductions (more than one launch's packed table of 208 jobs) of mixed shapes: every sum is the single-job
    kernel's, bit for bit."""
    from fastvim_amd import mixer_ops as M
    torch.manual_seed(3)
    shapes = [(6, 1024), (64, 352), (12, 96), (130, 132), (1, 40), (9, 260)]
    parts = [torch.randn(*shapes[k % len(shapes)], device="cuda") for k in range(260)]
    base = [torch.randn(p.shape[1], device="cuda") for p in parts]
    single = [b.clone() for b in base]
    for p, o in zip(parts, single):
        M.reduce_partials(p, p.shape[0], out=o, accumulate=True, defer=False)
    multi = [b.clone() for b in base]
    M.defer_reductions(True)
    try:
        for p, o in zip(parts, multi):
            M.reduce_partials(p, p.shape[0], out=o, accumulate=True)
        M.flush_reductions()
    finally:
        M.defer_reductions(False)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(single, multi))
