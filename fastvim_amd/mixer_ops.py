"""Thin typed wrappers over the fused-mixer C-ABI entry points (include/fastvim_hip.h).
No arithmetic happens here: allocate outputs with torch, pass pointers + sizes + stream."""
import ctypes

import torch

from . import _lib as L

f32 = ctypes.c_float


def _geo(rows, cols, transposed):
    # sequence position (i, j) -> memory token i*s_i + j*s_j  (see csrc/rowwalk.h)
    return (1, rows) if transposed else (cols, 1)


def conv_pool_fwd(xz, conv_w, conv_b, conv_w_b, conv_b_b, rows, cols, transposed, pool_max, scaling, tpp=1,
                  D=None, D_b=None):
    """Returns xc (2, B, rows*tpp, d_in); with D / D_b also skip (B, L, d_in) = D*conv_f + D_b*conv_b; with
    ``pool_max`` also amax (like xc: column of each maximum, for the backward pass)."""
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    xc = torch.empty(2, B, rows * tpp, d_in, device=xz.device, dtype=xz.dtype)
    skip = torch.empty(B, Ltok, d_in, device=xz.device, dtype=xz.dtype) if D is not None else None
    amax = torch.empty_like(xc) if pool_max else None
    rc = L.lib().fv_mixer_conv_pool_fwd(
        L.ptr(xz), L.ptr(conv_w), L.ptr(conv_b), L.ptr(conv_w_b), L.ptr(conv_b_b), L.ptr(D), L.ptr(D_b), L.ptr(xc),
        L.ptr(skip), L.ptr(amax), L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.i32(tpp), L.i32(d_in),
        L.i32(conv_w.shape[-1]), L.i32(pool_max), f32(scaling), L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_conv_pool_fwd")
    if pool_max:
        return (xc, amax) if D is None else (xc, skip, amax)
    return xc if D is None else (xc, skip)


def scan_fwd(xc, x_dbl, dt_w, dt_b, A_log, dt_w_b, dt_b_b, A_log_b, want_ckpt=False):
    """yc (2, B, Lc, d_in) fp32.  ``want_ckpt`` (training): returns (yc, ckpt) -- ckpt the state entering every 16-step
    chunk, which ``scan_bwd(..., ckpt=ckpt)`` takes instead of sweeping forward itself; None where the chunked backward
    kernel does not apply (short pooled lengths, dt_rank > 48)."""
    _, B, Lc, d_in = xc.shape
    R = dt_w.shape[1]
    N = A_log.shape[1]
    lib = L.lib()
    yc = torch.empty(2, B, Lc, d_in, device=xc.device, dtype=torch.float32)
    ckpt = None
    if want_ckpt:
        nck = lib.fv_mixer_scan_ckpt_floats(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(N), L.i32(R))
        ckpt = torch.empty(nck, device=xc.device, dtype=torch.float32) if nck else None
    # long sequences on few batch elements (un-pooled Vim at high resolution): segments of the sequence side by side
    nws = lib.fv_mixer_scan_fwd_seg_floats(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(N), L.i32(R))
    ws = torch.empty(nws, device=xc.device, dtype=torch.float32) if nws else None
    rc = lib.fv_mixer_scan_fwd_seg(
        L.ptr(xc), L.ptr(x_dbl), L.ptr(dt_w), L.ptr(dt_b), L.ptr(A_log), L.ptr(dt_w_b), L.ptr(dt_b_b),
        L.ptr(A_log_b), L.ptr(yc), L.ptr(ckpt), L.ptr(ws), L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(R), L.i32(N),
        L.i32(L.dtype_code(xc.dtype)), L.stream_of(xc))
    L.check(rc, "mixer_scan_fwd")
    return (yc, ckpt) if want_ckpt else yc


def xproj_scan_fwd(xc, Wx2_c, dt_w, dt_b, A_log, dt_w_b, dt_b_b, A_log_b):
    """x_proj + dt_proj + scan in one launch (short pooled lengths, bf16): returns (x_dbl (2, B*Lc, W) bf16, yc fp32),
    or None when the shape is not covered (the caller then runs xproj_fwd + scan_fwd)."""
    _, B, Lc, d_in = xc.shape
    R, N = dt_w.shape[1], A_log.shape[1]
    W = Wx2_c.shape[1]
    lib = L.lib()
    if (xc.dtype != torch.bfloat16 or Wx2_c.dtype != torch.bfloat16 or N != 16 or W != R + 2 * N
            or not xc.is_contiguous() or not Wx2_c.is_contiguous() or xc.data_ptr() % 16 or Wx2_c.data_ptr() % 16
            or not lib.fv_mixer_xproj_scan_fwd_ok(L.i32(Lc), L.i32(d_in), L.i32(R), L.i32(L.dtype_code(xc.dtype)))):
        return None
    x_dbl = torch.empty(2, B * Lc, W, device=xc.device, dtype=xc.dtype)
    yc = torch.empty(2, B, Lc, d_in, device=xc.device, dtype=torch.float32)
    rc = lib.fv_mixer_xproj_scan_fwd(
        L.ptr(xc), L.ptr(Wx2_c), L.ptr(dt_w), L.ptr(dt_b), L.ptr(A_log), L.ptr(dt_w_b), L.ptr(dt_b_b), L.ptr(A_log_b),
        L.ptr(x_dbl), L.ptr(yc), L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(R), L.i32(N), L.i32(L.dtype_code(xc.dtype)),
        L.stream_of(xc))
    L.check(rc, "mixer_xproj_scan_fwd")
    return x_dbl, yc


def combine_buffers(xz, ln_w):
    """The outputs of ``combine_fwd``, unwritten: g (B, L, d_in), mean, rstd (B*L) fp32 (None without a LayerNorm)."""
    B, Ltok, two_d = xz.shape
    g = torch.empty(B, Ltok, two_d // 2, device=xz.device, dtype=xz.dtype)
    if ln_w is not None:
        return g, torch.empty(B * Ltok, device=xz.device, dtype=torch.float32), torch.empty(B * Ltok, device=xz.device, dtype=torch.float32)
    return g, None, None


def combine_fwd(xz, skip, yc, ln_w, ln_b, eps, rows, cols, transposed, tpp=1, out=None):
    """g (B, L, d_in) = LayerNorm((yc_f + yc_b + skip) / 2) * silu(z); also returns mean, rstd (B*L) fp32.
    ``out``: (g, mean, rstd) buffers of ``combine_buffers`` to write instead of fresh ones."""
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    g, mean, rstd = out if out is not None else combine_buffers(xz, ln_w)
    rc = L.lib().fv_mixer_combine_fwd(
        L.ptr(xz), L.ptr(skip), L.ptr(yc), L.ptr(ln_w), L.ptr(ln_b), f32(eps), L.ptr(g), L.ptr(mean), L.ptr(rstd),
        L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.i32(tpp), L.i32(d_in),
        L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_combine_fwd")
    return g, mean, rstd


class _Deferred:
    """Gradient-partial reductions whose results are only needed before the optimizer step are queued
    (flat training state only) and issued up to 144 at a time by ONE launch (fv_reduce_partials_multi).
    ``side = True`` issues them on a second HIP stream (forked after the producers, joined in
    ``flush_reductions``; partial buffers stay referenced until the join): measured 3 % SLOWER under graph
    replay on MI355X (9.54 vs 9.28 ms/step), like the weight-gradient side stream, so it is off by default."""
    enabled = False
    max_jobs = 144       # per launch (the C side's table takes 208: one launch for a FastVim-T step's 197 jobs measured 5.216 ms
                         # against 5.209 with two, 5.237 with the three of a 96-job table -- profiles/r06_ab_reduce_one_launch.log)
    side = False         # second-stream issue: measured slower (see above)
    jobs = []
    stream = None
    pending = []

    @classmethod
    def add(cls, part, out, n_partials):
        lo, hi = out.data_ptr(), out.data_ptr() + out.numel() * 4
        if any(j[1].data_ptr() < hi and lo < j[1].data_ptr() + j[1].numel() * 4 for j in cls.jobs):
            # jobs of one launch read-modify-write their destinations concurrently: a second sum into the same
            # gradient (gradient accumulation) must wait for the first, or updates are lost nondeterministically
            cls.flush()
        cls.jobs.append((part, out, n_partials))
        if len(cls.jobs) >= cls.max_jobs:
            cls.flush()

    @classmethod
    def _launch(cls, jobs):
        k = len(jobs)
        ins = (ctypes.c_void_p * k)(*[j[0].data_ptr() for j in jobs])
        outs = (ctypes.c_void_p * k)(*[j[1].data_ptr() for j in jobs])
        Ss = (ctypes.c_int * k)(*[j[2] for j in jobs])
        ns = (ctypes.c_size_t * k)(*[j[0].numel() // j[2] for j in jobs])
        rc = L.lib().fv_reduce_partials_multi(ins, outs, Ss, ns, L.i32(k), L.i32(1), L.stream_of(jobs[0][0]))
        L.check(rc, "reduce_partials_multi")

    @classmethod
    def flush(cls):
        if not cls.jobs:
            return
        jobs, cls.jobs = cls.jobs, []
        if not cls.side:
            cls._launch(jobs)
            return
        cur = torch.cuda.current_stream()
        if cls.stream is None:
            cls.stream = torch.cuda.Stream()
        cls.stream.wait_stream(cur)
        with torch.cuda.stream(cls.stream):
            cls._launch(jobs)
        cls.pending.append(jobs)

    @classmethod
    def join(cls):
        if cls.pending:
            torch.cuda.current_stream().wait_stream(cls.stream)
            cls.pending = []


def defer_reductions(on):
    flush_reductions()
    _Deferred.enabled = bool(on)


def pending_reductions():
    return len(_Deferred.jobs)


def drop_reductions():
    """Forget queued (not yet issued) reductions; returns how many were dropped."""
    n = len(_Deferred.jobs)
    _Deferred.jobs = []
    return n


def flush_reductions():
    """Issue whatever is queued and make the current stream wait for every outstanding reduction."""
    _Deferred.flush()
    _Deferred.join()


def reduce_partials(part, n_partials, out=None, accumulate=False, defer=True):
    """Fixed-order sum over the leading dim of a (n_partials, ...) fp32 buffer.  ``out`` (contiguous
    fp32, same element count) receives the result; ``accumulate`` adds into it instead (and may be
    deferred, see _Deferred)."""
    if out is not None and accumulate and defer and _Deferred.enabled:
        assert out.numel() == part.numel() // n_partials and out.is_contiguous() and out.dtype == torch.float32
        _Deferred.add(part, out, n_partials)
        return out
    if out is None:
        out = torch.empty(part.shape[1:], device=part.device, dtype=torch.float32)
    n = part.numel() // n_partials
    assert out.numel() == n and out.is_contiguous() and out.dtype == torch.float32
    rc = L.lib().fv_reduce_partials(L.ptr(part), L.ptr(out), L.i32(n_partials), ctypes.c_size_t(n),
                                    L.i32(accumulate), L.stream_of(part))
    L.check(rc, "reduce_partials")
    return out


def combine_out_proj_addnorm_ok(xz, rows, cols, tpp, d_model):
    B, Ltok, two_d = xz.shape
    return bool(xz.dtype == torch.bfloat16 and xz.is_contiguous() and
                L.lib().fv_mixer_combine_out_proj_addnorm_ok(L.i32(B), L.i32(rows), L.i32(cols), L.i32(tpp), L.i32(two_d // 2),
                                                             L.i32(d_model), L.i32(L.FV_BF16)))


def combine_out_proj_addnorm(xz, skip, yc, ln_w, ln_b, ln_eps, rows, cols, transposed, out, W_out_c, residual2, norm_w32,
                             row_scale, rows_per_scale, eps):
    """``combine_fwd`` into the buffers ``out`` = (g, mean, rstd) AND out_proj + DropPath scale + residual add + RMSNorm
    of the result (``fv_gemm_bf16_addnorm``) in one launch; returns (normed (M, d) bf16, residual_out (M, d) fp32,
    rstd (M))."""
    B, Ltok, two_d = xz.shape
    d = W_out_c.shape[0]
    Mrows = B * Ltok
    s_i, s_j = _geo(rows, cols, transposed)
    g, mean, rstd_ln = out
    y = torch.empty(Mrows, d, device=xz.device, dtype=torch.bfloat16)
    res_out = torch.empty(Mrows, d, device=xz.device, dtype=torch.float32)
    rstd = torch.empty(Mrows, device=xz.device, dtype=torch.float32)
    rc = L.lib().fv_mixer_combine_out_proj_addnorm(
        L.ptr(xz), L.ptr(skip), L.ptr(yc), L.ptr(ln_w), L.ptr(ln_b), f32(ln_eps), L.ptr(g), L.ptr(mean), L.ptr(rstd_ln),
        L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.ptr(W_out_c), ctypes.c_long(W_out_c.stride(0)),
        L.ptr(residual2), L.ptr(norm_w32), L.ptr(row_scale), L.i32(rows_per_scale), L.ptr(y), L.ptr(res_out), L.ptr(rstd),
        f32(eps), L.stream_of(xz))
    L.check(rc, "mixer_combine_out_proj_addnorm")
    return y, res_out, rstd


def combine_bwd(dg, xz, skip, yc, ln_w, ln_b, mean, rstd, dxz, rows, cols, transposed, grad_out=None, tpp=1):
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    lib = L.lib()
    nb = lib.fv_mixer_bwd_blocks(L.i32(B), L.i32(rows), L.i32(d_in), L.i32(tpp), L.i32(0))
    d_o = torch.empty(B, Ltok, d_in, device=xz.device, dtype=xz.dtype)
    dyc = torch.empty(B, rows * tpp, d_in, device=xz.device, dtype=torch.float32)
    part = torch.empty(nb, 2, d_in, device=xz.device, dtype=torch.float32)
    rc = lib.fv_mixer_combine_bwd(
        L.ptr(dg), L.ptr(xz), L.ptr(skip), L.ptr(yc), L.ptr(ln_w), L.ptr(ln_b), L.ptr(mean), L.ptr(rstd), L.ptr(dxz), L.ptr(d_o),
        L.ptr(dyc), L.ptr(part), L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.i32(tpp), L.i32(d_in),
        L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_combine_bwd")
    if grad_out is not None:
        reduce_partials(part, nb, out=grad_out, accumulate=True)
        return d_o, dyc, None
    return d_o, dyc, reduce_partials(part, nb)      # (2, d_in): [dln_w, dln_b]


def scan_bwd(xc, x_dbl, dt_w, dt_b, A_log, dt_w_b, dt_b_b, A_log_b, dyc, grad_out=None, keep_chunks=False,
             dyc_per_direction=False, ckpt=None):
    """Returns (dxc, dx_dbl, pr) with pr (2, d_in*(N+R+1)) = per direction [dA_log | d dt_w | d dt_bias];
    when ``grad_out`` (flat fp32 view of exactly that layout) is given, the sums are accumulated into
    it instead and pr is None.  ``dyc_per_direction``: dyc is (2, B, Lc, d_in) instead of one (B, Lc, d_in)
    shared by both directions (the MAE masked mixer)."""
    _, B, Lc, d_in = xc.shape
    assert dyc.numel() == (2 if dyc_per_direction else 1) * B * Lc * d_in and dyc.dtype == torch.float32
    R, N = dt_w.shape[1], A_log.shape[1]
    W = R + 2 * N
    dev = xc.device
    lib = L.lib()
    f32o = dict(device=dev, dtype=torch.float32)
    given = ckpt is not None          # checkpoints of the forward launch (scan_fwd(want_ckpt=True))
    # long sequences on few batch elements, with the forward launch's checkpoints: the segments side by side (one partial
    # row of parameter gradients per batch element and segment).  Segment count, channel chunks and partial rows are all
    # asked for the REAL d_inner (an explicit dt_rank / expand != 2 makes it differ from 32 dt_rank)
    nws = lib.fv_mixer_scan_bwd_seg_floats(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(N), L.i32(R)) if given else 0
    nchunks = lib.fv_mixer_scan_bwd_seg_chunks(L.i32(B), L.i32(d_in), L.i32(Lc), L.i32(R), L.i32(int(nws > 0)))
    dxc = torch.empty(2, B, Lc, d_in, **f32o)
    dx_dbl = torch.empty(nchunks, 2, B * Lc, W, **f32o)
    if not given:
        nck = lib.fv_mixer_scan_bwd_ckpt_floats(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(N), L.i32(R))
        ckpt = torch.empty(nck, **f32o) if nck else None
    ws = torch.empty(nws, **f32o) if nws else None
    nprt = lib.fv_mixer_scan_bwd_seg_partials(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(R)) if nws else \
        lib.fv_mixer_scan_bwd_partials(L.i32(B), L.i32(Lc), L.i32(R))
    part = torch.empty(nprt, 2 * d_in * (N + R + 1), **f32o)
    rc = lib.fv_mixer_scan_bwd_seg(
        L.ptr(xc), L.ptr(x_dbl), L.ptr(dt_w), L.ptr(dt_b), L.ptr(A_log), L.ptr(dt_w_b), L.ptr(dt_b_b),
        L.ptr(A_log_b), L.ptr(dyc), L.i32(int(dyc_per_direction)), L.ptr(dxc), L.ptr(dx_dbl), L.ptr(ckpt), L.i32(int(given)),
        L.ptr(part), L.ptr(ws), L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(R), L.i32(N), L.i32(L.dtype_code(xc.dtype)),
        L.stream_of(xc))
    L.check(rc, "mixer_scan_bwd")
    if not keep_chunks:       # keep_chunks: the x_proj adjoint kernel sums the chunk partials itself
        dx_dbl = dx_dbl[0] if nchunks == 1 else reduce_partials(dx_dbl, nchunks)
    if grad_out is not None:
        reduce_partials(part, nprt, out=grad_out, accumulate=True)
        return dxc, dx_dbl, None
    return dxc, dx_dbl, reduce_partials(part, nprt).view(2, d_in * (N + R + 1))


def scan_bwd_xproj_ok(xc, dt_w, pool_max, rows, cols, tpp):
    """Is the short backward scan with the x_proj adjoint folded in built for this shape (and does the conv + pool
    adjoint that consumes its two-part pooled gradient take it)?"""
    _, B, Lc, d_in = xc.shape
    lib = L.lib()
    return bool(xc.dtype in (torch.float32, torch.bfloat16)
                and lib.fv_mixer_scan_bwd_xproj_ok(L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(dt_w.shape[1]), L.i32(L.dtype_code(xc.dtype)))
                and lib.fv_mixer_conv_pool_bwd2_ok(L.i32(rows), L.i32(cols), L.i32(tpp), L.i32(d_in), L.i32(int(bool(pool_max)))))


def scan_bwd_xproj(xc, x_dbl, dt_w, dt_b, A_log, dt_w_b, dt_b_b, A_log_b, dyc, Wx, Wx_b, grad_out=None, Wx2_bf16=None):
    """The short backward scan with the x_proj adjoint's data half folded in (fv_mixer_scan_bwd_xproj): returns
    (dxc, dxc2, dx_dbl_chunks, pr).  dxc (2, B, Lc, d_in) fp32 + dxc2 (same shape, storage dtype) is the TOTAL gradient of
    the pooled conv output (through the scan and through x_proj); ``conv_pool_bwd(..., dxc2=dxc2)`` adds the two.
    dx_dbl_chunks (2, 2, B * Lc, W) fp32: the per-chunk partial rows (their sum is the x_proj weight gradient's operand)."""
    _, B, Lc, d_in = xc.shape
    R, N = dt_w.shape[1], A_log.shape[1]
    W = R + 2 * N
    lib = L.lib()
    f32o = dict(device=xc.device, dtype=torch.float32)
    dxc = torch.empty(2, B, Lc, d_in, **f32o)
    dxc2 = torch.empty(2, B, Lc, d_in, device=xc.device, dtype=xc.dtype)
    dx_dbl = torch.empty(2, 2, B * Lc, W, **f32o)
    nprt = lib.fv_mixer_scan_bwd_partials(L.i32(B), L.i32(Lc), L.i32(R))
    part = torch.empty(nprt, 2 * d_in * (N + R + 1), **f32o)
    if xc.dtype == torch.bfloat16 and Wx2_bf16 is None:      # (2, W, d_in) bf16: the shadow weights where the caller has them
        Wx2_bf16 = torch.stack([Wx, Wx_b]).to(torch.bfloat16)
    assert Wx2_bf16 is None or (Wx2_bf16.dtype == torch.bfloat16 and Wx2_bf16.is_contiguous() and Wx2_bf16.shape == (2, W, d_in))
    rc = lib.fv_mixer_scan_bwd_xproj(
        L.ptr(xc), L.ptr(x_dbl), L.ptr(dt_w), L.ptr(dt_b), L.ptr(A_log), L.ptr(dt_w_b), L.ptr(dt_b_b), L.ptr(A_log_b),
        L.ptr(dyc), L.ptr(Wx), L.ptr(Wx_b), L.ptr(Wx2_bf16), L.ptr(dxc), L.ptr(dxc2), L.ptr(dx_dbl), L.ptr(part), L.i32(B), L.i32(Lc),
        L.i32(d_in), L.i32(R), L.i32(N), L.i32(L.dtype_code(xc.dtype)), L.stream_of(xc))
    L.check(rc, "mixer_scan_bwd_xproj")
    if grad_out is not None:
        reduce_partials(part, nprt, out=grad_out, accumulate=True)
        return dxc, dxc2, dx_dbl, None
    return dxc, dxc2, dx_dbl, reduce_partials(part, nprt).view(2, d_in * (N + R + 1))


def chunk_rows_bf16(jobs):
    """jobs: [(dx_dbl_chunks (nchunks, 2, M, W) fp32, out (2, M, WP) bf16)]: out = bf16(sum over the chunks), pad columns
    zero -- ONE launch per job shape for up to 64 mixers (fv_chunk_rows_bf16).  The queue of a backward pass holds every
    block's job: on a non-square grid (224 x 256 px: 14 pooled rows on even layers, 16 on the rotated ones) or with
    micro-batches of different sizes the shapes differ, so jobs are bucketed by shape, in queue order inside a bucket."""
    lib = L.lib()
    buckets = {}
    for j in jobs:
        assert j[1].dtype == torch.bfloat16 and j[1].is_contiguous() and j[0].dtype == torch.float32 and j[0].is_contiguous()
        buckets.setdefault((tuple(j[0].shape), tuple(j[1].shape)), []).append(j)
    for (ishape, _), bjobs in buckets.items():
        nchunks, _, Mrows, W = ishape
        for lo in range(0, len(bjobs), 64):
            grp = bjobs[lo:lo + 64]
            k = len(grp)
            ins = (ctypes.c_void_p * k)(*[j[0].data_ptr() for j in grp])
            outs = (ctypes.c_void_p * k)(*[j[1].data_ptr() for j in grp])
            rc = lib.fv_chunk_rows_bf16(ins, outs, L.i32(k), L.i32(nchunks), ctypes.c_long(2 * Mrows), L.i32(W), L.stream_of(grp[0][0]))
            L.check(rc, "chunk_rows_bf16")


def conv_pool_bwd(xz, d_o, dxc, conv_w, conv_b, conv_w_b, conv_b_b, D, D_b, dxz, rows, cols, transposed,
                  pool_max, scaling, grad_out=None, tpp=1, amax=None, dxc2=None):
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    lib = L.lib()
    nb = lib.fv_mixer_bwd_blocks(L.i32(B), L.i32(rows), L.i32(d_in), L.i32(tpp), L.i32(1))
    part = torch.empty(nb, 12 * d_in, device=xz.device, dtype=torch.float32)
    assert dxc2 is None or (dxc2.dtype == xz.dtype and dxc2.shape == dxc.shape and dxc2.is_contiguous())
    rc = lib.fv_mixer_conv_pool_bwd2(
        L.ptr(xz), L.ptr(d_o), L.ptr(dxc), L.ptr(dxc2), L.ptr(conv_w), L.ptr(conv_b), L.ptr(conv_w_b), L.ptr(conv_b_b),
        L.ptr(D), L.ptr(D_b), L.ptr(amax), L.ptr(dxz), L.ptr(part), L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i),
        L.i32(s_j), L.i32(tpp), L.i32(d_in), L.i32(conv_w.shape[-1]), L.i32(pool_max), f32(scaling),
        L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_conv_pool_bwd")
    # segments: [dw (d_in*4) | dw_b (d_in*4) | db | db_b | dD | dD_b]
    if grad_out is not None:
        reduce_partials(part, nb, out=grad_out, accumulate=True)
        return None
    return reduce_partials(part, nb)


def conv_pool_bwd_dgrad_ok(xz, rows, cols, tpp, d_model, pool_max):
    """Is the conv + pool adjoint built as the A-tile producer of the in_proj data gradient + norm adjoint for this shape?"""
    B, Ltok, two_d = xz.shape
    return bool(xz.dtype == torch.bfloat16 and xz.is_contiguous() and
                L.lib().fv_mixer_conv_pool_bwd_dgrad_ok(L.i32(B), L.i32(rows), L.i32(cols), L.i32(tpp), L.i32(two_d // 2),
                                                        L.i32(d_model), L.i32(int(bool(pool_max))), L.i32(L.FV_BF16)))


def conv_pool_bwd_dgrad(xz, d_o, dxc, dxc2, conv_w, conv_b, conv_w_b, conv_b_b, D, D_b, dxz, rows, cols, transposed, scaling,
                        W_in_t, dres_out, r, rstd, norm_w32, row_scale, rows_per_scale, W2=None, conv_grad_out=None):
    """``conv_pool_bwd`` (x half of ``dxz`` written, its z half read) AND ``fv_gemm_bf16_dgrad_addnorm_bwd2`` on the result in
    one launch (fv_mixer_conv_pool_bwd_dgrad).  ``W_in_t`` (d, 2 d_in) bf16 = in_proj.weight^T.  Returns
    (conv parameter-gradient sums or None when accumulated into ``conv_grad_out``, dx (M, d) bf16, dres_in (M, d) fp32,
    pw (nb, d) partial sums of the norm weight's gradient, nb, dg_prev (M, N2) bf16 or None)."""
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    d = W_in_t.shape[0]
    Mrows = B * Ltok
    s_i, s_j = _geo(rows, cols, transposed)
    lib = L.lib()
    dev = xz.device
    nb = lib.fv_mixer_conv_pool_bwd_dgrad_blocks(L.i32(B), L.i32(rows))
    part = torch.empty(nb, 12 * d_in, device=dev, dtype=torch.float32)
    dx = torch.empty(Mrows, d, device=dev, dtype=torch.bfloat16)
    dres_in = torch.empty(Mrows, d, device=dev, dtype=torch.float32)
    pw = torch.empty(nb, d, device=dev, dtype=torch.float32)
    N2 = W2.shape[1] if W2 is not None else 0
    dg_prev = torch.empty(Mrows, N2, device=dev, dtype=torch.bfloat16) if W2 is not None else None
    assert dxc2 is None or (dxc2.dtype == xz.dtype and dxc2.shape == dxc.shape and dxc2.is_contiguous())
    assert W_in_t.dtype == torch.bfloat16 and W_in_t.stride(1) == 1 and W_in_t.shape[1] == two_d
    rc = lib.fv_mixer_conv_pool_bwd_dgrad(
        L.ptr(xz), L.ptr(d_o), L.ptr(dxc), L.ptr(dxc2), L.ptr(conv_w), L.ptr(conv_b), L.ptr(conv_w_b), L.ptr(conv_b_b),
        L.ptr(D), L.ptr(D_b), L.ptr(dxz), L.ptr(part), L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j),
        f32(scaling), L.ptr(W_in_t), ctypes.c_long(W_in_t.stride(0)), L.ptr(dres_out), L.ptr(r), L.ptr(rstd), L.ptr(norm_w32),
        L.ptr(row_scale), L.i32(rows_per_scale), L.ptr(dx), L.ptr(dres_in), L.ptr(pw), L.ptr(W2), L.ptr(dg_prev), L.i32(N2),
        ctypes.c_long(W2.stride(0) if W2 is not None else 0), L.stream_of(xz))
    L.check(rc, "mixer_conv_pool_bwd_dgrad")
    if conv_grad_out is not None:
        reduce_partials(part, nb, out=conv_grad_out, accumulate=True)
        p2 = None
    else:
        p2 = reduce_partials(part, nb)
    return p2, dx, dres_in, pw, nb, dg_prev


def transpose_bf16_batched(srcs, dsts):
    """dsts[j] (cols, rows) = srcs[j] (rows, cols)^T, bf16, equal shapes, one launch per 64 matrices."""
    lib = L.lib()
    rows, cols = srcs[0].shape
    for lo in range(0, len(srcs), 64):
        a, b = srcs[lo:lo + 64], dsts[lo:lo + 64]
        assert all(t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape) == (rows, cols) for t in a)
        assert all(t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape) == (cols, rows) for t in b)
        k = len(a)
        ins = (ctypes.c_void_p * k)(*[t.data_ptr() for t in a])
        outs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in b])
        L.check(lib.fv_transpose_bf16_batched(ins, outs, L.i32(k), L.i32(rows), L.i32(cols), L.stream_of(a[0])), "transpose_bf16_batched")


XPROJ_WIDTHS = (44, 56, 80, 96, 112, 34, 36, 38, 64)


def xproj_fwd(xc, Wx2_c):
    """x_dbl (2, M, W) = xc (2, B, Lc, d_in) @ Wx2_c (2, W, d_in)^T, both directions in one launch (bf16 operands,
    fp32 accumulate).  The LDS-free kernel wins where the weight is small (FastVim-T: 4.0 vs 4.9-5.2 us for
    hipBLASLt) and loses for wider models, where every 16-row block re-reads a 100+ KB weight from L2 (FastVim-B:
    11.4 vs 7.7 us) -- fp32 and odd shapes go to the fp32-MFMA GEMM, both directions as its batch dimension."""
    d_in = xc.shape[-1]
    Mrows = xc.numel() // (2 * d_in)
    W = Wx2_c.shape[1]
    if (xc.dtype != torch.bfloat16 or Wx2_c.dtype != torch.bfloat16 or d_in % 32 or W > 112
            or not xc.is_contiguous() or not Wx2_c.is_contiguous() or xc.data_ptr() % 16 or Wx2_c.data_ptr() % 16):
        if xc.is_cuda and xc.dtype in (torch.float32, torch.bfloat16) and Wx2_c.dtype in (torch.float32, torch.bfloat16):
            from .gemm import gemm_any_bnt
            return gemm_any_bnt(xc.reshape(2, Mrows, d_in), Wx2_c)
        return torch.bmm(xc.view(2, Mrows, d_in), Wx2_c.transpose(1, 2))
    out = torch.empty(2, Mrows, W, device=xc.device, dtype=xc.dtype)
    rc = L.lib().fv_mixer_xproj_fwd(L.ptr(xc), L.ptr(Wx2_c), L.ptr(out), L.i32(Mrows), L.i32(d_in), L.i32(W), L.stream_of(xc))
    L.check(rc, "mixer_xproj_fwd")
    return out


_XPROJ_PRESUM = 8       # chunk count from which dx_dbl is summed by the reduction kernel first (FastVim-B: 8 chunks of 192
                        # channels; same box 30.85 -> 30.78 ms per step, twice; 16 in round 2)


def xproj_bwd3_ok(Mrows, d_in, W, dtype):
    return bool(dtype == torch.bfloat16 and L.lib().fv_mixer_xproj_bwd3_ok(L.i32(Mrows), L.i32(d_in), L.i32(W), L.i32(L.dtype_code(dtype))))


def conv_pool_bwd2_ok(rows, cols, tpp, d_in, pool_max):
    """The conv + pool adjoint takes the pooled gradient as two addends (fp32 dxc + storage-dtype dxc2) at this shape."""
    return bool(L.lib().fv_mixer_conv_pool_bwd2_ok(L.i32(rows), L.i32(cols), L.i32(tpp), L.i32(d_in), L.i32(int(pool_max))))


def xproj_bwd(dx_dbl_chunks, xc, Wx, Wx_b, dxc, grad_out=None, dw=True, Wx2_t=None, dxc2=None):
    """dxc (2, B, Lc, d_in) fp32 += dx_dbl @ Wx (in place); returns d x_proj weights (2, W, d_in) fp32, or
    accumulates them into ``grad_out`` (flat view of that shape) and returns None.  ``dw=False``: the weight gradient
    is left to the caller (grouped GEMM); returns the summed dx_dbl rows as bf16, (2, M, W rounded up to 8) with zero
    pad columns.  ``Wx2_t`` (2, d_in, W) bf16, with ``dw=False``: the transposed compute-dtype shadow of both weights --
    the product then runs on the bf16 matrix cores where that form is built (fv_mixer_xproj_bwd3); with ``dxc2`` (bf16,
    dxc's shape; only where ``xproj_bwd3_ok``) it is WRITTEN there and dxc is left alone."""
    nchunks, _, Mrows, W = dx_dbl_chunks.shape
    d_in = xc.shape[-1]
    lib = L.lib()
    # every 128-channel block of the adjoint kernel re-sums the chunk partials of its rows: nchunks x d_in / 128 reads
    # of each row (8 x 12 at FastVim-B with the short kernel's 192-channel chunks).  From 8 chunks up they are summed once, by the reduction
    # kernel: FastVim-B 37.2 -> 36.6 ms per step; neutral to slightly worse at 12 chunks (FastVim-S), so not there.
    if nchunks >= _XPROJ_PRESUM:
        dx_dbl_chunks = reduce_partials(dx_dbl_chunks, nchunks, defer=False).view(1, 2, Mrows, W)
        nchunks = 1
    if not dw:
        WP = (W + 7) // 8 * 8
        dxb = torch.empty(2, Mrows, WP, device=xc.device, dtype=torch.bfloat16)
        if Wx2_t is not None:
            assert Wx2_t.dtype == torch.bfloat16 and Wx2_t.is_contiguous() and tuple(Wx2_t.shape) == (2, d_in, W)
            assert dxc2 is None or (dxc2.dtype == torch.bfloat16 and dxc2.is_contiguous() and dxc2.numel() == dxc.numel())
            rc = lib.fv_mixer_xproj_bwd3(L.ptr(dx_dbl_chunks), L.i32(nchunks), L.ptr(xc), L.ptr(Wx), L.ptr(Wx_b), L.ptr(Wx2_t),
                                         L.ptr(dxc), L.ptr(dxc2), L.ptr(dxb), L.i32(Mrows), L.i32(d_in), L.i32(W),
                                         L.i32(L.dtype_code(xc.dtype)), L.stream_of(xc))
        else:
            assert dxc2 is None
            rc = lib.fv_mixer_xproj_bwd2(L.ptr(dx_dbl_chunks), L.i32(nchunks), L.ptr(xc), L.ptr(Wx), L.ptr(Wx_b), L.ptr(dxc),
                                         None, L.ptr(dxb), L.i32(Mrows), L.i32(d_in), L.i32(W),
                                         L.i32(L.dtype_code(xc.dtype)), L.stream_of(xc))
        L.check(rc, "mixer_xproj_bwd")
        return dxb
    ns = lib.fv_mixer_xproj_bwd_slices(L.i32(Mrows))
    part = torch.empty(ns, 2, W, d_in, device=xc.device, dtype=torch.float32)
    rc = lib.fv_mixer_xproj_bwd(L.ptr(dx_dbl_chunks), L.i32(nchunks), L.ptr(xc), L.ptr(Wx), L.ptr(Wx_b), L.ptr(dxc),
                                L.ptr(part), L.i32(Mrows), L.i32(d_in), L.i32(W), L.i32(L.dtype_code(xc.dtype)),
                                L.stream_of(xc))
    L.check(rc, "mixer_xproj_bwd")
    if grad_out is not None:
        reduce_partials(part, ns, out=grad_out.view(-1), accumulate=True)
        return None
    return reduce_partials(part, ns)


def rows_segment_sum(x, idx, rows, scale=1.0, out_dtype=None):
    """out (2, B, rows, d) = scale * sum of the tokens of each pooling row, per direction.  x: (2, B, Lk, d) or
    (B, Lk, d) shared by both directions; idx (2, B, Lk) int32 = pooling row of token t per direction."""
    per_dir = x.dim() == 4
    B, Lk, d = x.shape[-3:]
    assert idx.shape == (2, B, Lk) and idx.dtype == torch.int32 and idx.is_contiguous() and x.is_contiguous()
    out = torch.empty(2, B, rows, d, device=x.device, dtype=out_dtype or x.dtype)
    rc = L.lib().fv_rows_segment_sum(L.ptr(x), L.i32(L.dtype_code(x.dtype)), L.i32(int(per_dir)), L.ptr(idx), L.ptr(out),
                                     L.i32(L.dtype_code(out.dtype)), L.i32(B), L.i32(Lk), L.i32(rows), L.i32(d),
                                     f32(scale), L.stream_of(x))
    L.check(rc, "rows_segment_sum")
    return out


def rows_gather(x, idx, scale=1.0, out_dtype=None):
    """out (2, B, Lk, d)[dir, b, t] = scale * x[dir, b, idx[dir, b, t]].  x: (2, B, rows, d)."""
    _, B, rows, d = x.shape
    Lk = idx.shape[-1]
    assert idx.shape == (2, B, Lk) and idx.dtype == torch.int32 and idx.is_contiguous() and x.is_contiguous()
    out = torch.empty(2, B, Lk, d, device=x.device, dtype=out_dtype or x.dtype)
    rc = L.lib().fv_rows_gather(L.ptr(x), L.i32(L.dtype_code(x.dtype)), L.ptr(idx), L.ptr(out),
                                L.i32(L.dtype_code(out.dtype)), L.i32(B), L.i32(Lk), L.i32(rows), L.i32(d), f32(scale),
                                L.stream_of(x))
    L.check(rc, "rows_gather")
    return out
