"""FastVim mixer module: mirror of ``Mamba`` in
mamba-1p1p1/mamba_ssm/modules/mamba_simple_faster.py:27-457 -- same constructor signature,
parameter names/shapes/initialisation (so reference ``state_dict``s load unchanged), same
``forward(hidden_states (B, L, D)) -> (B, L, D)``.

The body is not the reference's ~35-launch op chain: one autograd.Function drives the fused
channel-last HIP kernels (csrc/mixer_fwd.hip, csrc/mixer_bwd.hip) with a hand-written
backward; activations stay token-major so in_proj/out_proj are plain row-major GEMMs, and the
odd-layer grid transpose of ``Block`` is a stride pair, not a copy (``transposed_grid``).
"""
import ctypes
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import mixer_ops as M
from .gemm import gemm_any_bnn, gemm_any_btn, gemm_any_nn, gemm_any_nt, gemm_any_tn, gemm_nn, gemm_nt, gemm_tn


def _shadow(w, cdt):
    """bf16 shadow copy kept by FlatTrainingState, else a cast.  The shadow is refreshed by the fused optimizer
    kernel; any OTHER in-place write to the parameter (``load_state_dict``, an EMA copy-in, a torch optimizer) bumps
    the tensor's version counter, which is compared here, so a stale shadow is re-cast before it is used."""
    sh = getattr(w, "_fv_shadow", None)
    if sh is not None and sh.dtype == cdt:
        if w._version != w._fv_shadow_version:
            with torch.no_grad():
                sh.copy_(w)
            w._fv_shadow_version = w._version
        return sh
    return w.to(cdt)


def _shadow_t(w, cdt):
    """The TRANSPOSED compute-dtype copy of a 2-D weight (in_proj.weight (2 d_inner, d) -> (d, 2 d_inner), K-contiguous for
    the data gradient that streams it into MFMA registers: fv_mixer_conv_pool_bwd_dgrad).  FlatTrainingState keeps one per
    eligible in_proj and re-transposes them in one launch after every fused optimizer step (``refresh_transposed``); any
    other in-place write to the parameter is caught by the version counter, like ``_shadow``.  Without a flat training
    state: a fresh transpose."""
    wt = getattr(w, "_fv_shadow_t", None)
    if wt is not None and wt.dtype == cdt:
        if w._version != w._fv_shadow_t_version:
            with torch.no_grad():
                wt.copy_(_shadow(w, cdt).t())
            w._fv_shadow_t_version = w._version
        return wt
    return _shadow(w, cdt).t().contiguous()


def _wx2_shadow_t(fv):
    """The transposed bf16 shadow (2, d_inner, W) of a mixer's x_proj weight pair kept by FlatTrainingState for the wide
    models, or None.  Re-made by the flat state after every optimizer step; an in-place write to either weight since then
    (version counters) is caught here and the pair re-transposed from the (already re-cast) plain shadow."""
    if fv is None:
        return None
    wt = fv.get("Wx2_shadow_t")
    if wt is None:
        return None
    params = fv["Wx2_t_params"]
    if any(w._version != w._fv_shadow_t_version for w in params):
        with torch.no_grad():
            for k, w in enumerate(params):
                wt[k].copy_(_shadow(w, wt.dtype).t())
                w._fv_shadow_t_version = w._version
    return wt


def _direct_grad(w):
    """The preallocated .grad view my kernels may accumulate into (flat training state), or None."""
    g = w.grad
    if getattr(w, "_fv_direct", False) and g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g
    return None


def _wgrad(X, Y, W=None, splits=16):
    """dW (I, J) fp32 = X^T Y for X (M, I), Y (M, J) with M >> I, J: the reduction dim is split into
    `splits` batched GEMMs (enough workgroups to fill 256 CUs) whose fp32 partials are summed in a
    fixed order by fv_reduce_partials -- a deterministic split-K.  If the weight ``W`` owns a
    preallocated flat gradient, the sum is accumulated straight into it and None is returned."""
    M_, I = X.shape
    J = Y.shape[1]
    while splits > 1 and (M_ % splits or M_ // splits < 256):
        splits //= 2
    # fp32 operands, and bf16 ones whose token count is not a multiple of 64 (the head at batch 8): the fp32-MFMA kernel,
    # the K slices as its batch dimension
    part = gemm_any_tn(X, Y, splits)
    g = _direct_grad(W) if W is not None else None
    if g is not None:
        M.reduce_partials(part, splits, out=g.view(-1), accumulate=True)
        return None
    return M.reduce_partials(part, splits)


def _mfma_ok(*ts):
    return all(t.dtype == torch.bfloat16 and t.stride(-1) == 1 and t.data_ptr() % 16 == 0 for t in ts)


def _any_ok(*ts):
    return all(t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.dim() == 2 for t in ts)


def linear_fwd(a2, w_c, bias=None):
    """a2 (M, K) x w_c (N, K)^T (+ bias) in the compute dtype: the bf16 MFMA GEMM where its alignment rules hold, the
    fp32-MFMA GEMM otherwise (fp32 -- the reference's default precision -- and odd bf16 shapes): no library GEMM."""
    if _mfma_ok(a2, w_c) and a2.shape[1] % 8 == 0 and w_c.shape[0] % 8 == 0:      # 16-byte rows of A, W and C
        return gemm_nt(a2, w_c, bias=None if bias is None else bias.float())
    if _any_ok(a2, w_c):
        return gemm_any_nt(a2, w_c, bias)
    return F.linear(a2, w_c, None if bias is None else bias.to(a2.dtype))      # fp16 / fp64 / CPU callers of the op-level API


def linear_dgrad(g2, w_c):
    """g2 (M, N) x w_c (N, K) -> (M, K): data gradient with the weight as stored."""
    if _mfma_ok(g2, w_c) and w_c.shape[1] % 8 == 0 and g2.shape[1] % 8 == 0:
        return gemm_nn(g2, w_c)
    if _any_ok(g2, w_c):
        return gemm_any_nn(g2, w_c)
    return g2 @ w_c


class _SideStream:
    """Weight-gradient GEMMs are off the critical path of backward: in flat-training mode they are
    enqueued on a second HIP stream (fork after the producer, join in finish_backward) so they overlap
    the latency-bound row-walker / scan kernels of the main chain.  Operands are kept referenced
    until the join, so the caching allocator cannot hand their memory to main-stream kernels."""
    enabled = False
    stream = None
    pending = []

    @classmethod
    def run(cls, fn, *keep):
        if not cls.enabled:
            return fn()
        cur = torch.cuda.current_stream()
        if cls.stream is None:
            cls.stream = torch.cuda.Stream()
        cls.stream.wait_stream(cur)
        with torch.cuda.stream(cls.stream):
            out = fn()
        cls.pending.append(keep)
        return out

    @classmethod
    def join(cls):
        if cls.stream is not None and cls.pending:
            torch.cuda.current_stream().wait_stream(cls.stream)
        cls.pending = []


XPROJ_IN_SCAN = True      # the x_proj adjoint's data half inside the short scan backward where it is built (A/B switch)


class _GroupedWgrad:
    """Weight gradients are only needed before the optimizer step.  With the flat training state they are queued
    (operands kept alive) and computed by grouped launches (fv_gemm_bf16_tn_grouped, up to 40 problems each) at the end of
    the backward pass: one weight-gradient GEMM at FastVim-T is 168-336 workgroups, a fraction of what the chip holds
    at once, so each separate launch pays a tail; the grouped queue does not, and with the tail gone a smaller
    split-K factor (less fp32 partial traffic) is affordable."""
    enabled = False
    jobs = []
    # chunk > 0: a group is launched as soon as it holds that many problems (on the weight-gradient stream when there is
    # one) instead of after the last block.  Measured slower at FastVim-T (flat.py), so off.
    chunk = 0

    @classmethod
    def add(cls, g2, a2, out):
        from .gemm import grouped_splits
        if any(j[2].data_ptr() == out.data_ptr() for j in cls.jobs):
            # a second gradient for the same weight (gradient accumulation: two backward passes before one
            # finish_backward): problems of one grouped launch are summed into ``out`` concurrently, so the
            # earlier one is issued first -- same-destination sums stay ordered, i.e. deterministic
            cls.flush()
        cls.jobs.append((g2, a2, out, grouped_splits(g2.shape[0], M=g2.shape[1], N=a2.shape[1])))
        if cls.chunk and len(cls.jobs) >= cls.chunk:
            cls.flush()

    sums = []            # (partials, splits, out) of groups launched on the weight-gradient stream, not yet summed
    # operands that still have to be MADE before the group runs: (d x_dbl chunk partials fp32, bf16 rows) of the mixers
    # whose x_proj adjoint ran inside the scan backward (fv_mixer_scan_bwd_xproj) -- one launch for all of them
    rows_jobs = []

    @classmethod
    def flush(cls):
        if cls.rows_jobs:
            rj, cls.rows_jobs = cls.rows_jobs, []
            M.chunk_rows_bf16(rj)
        if cls.jobs:
            from .gemm import gemm_tn_grouped
            jobs, cls.jobs = cls.jobs, []
            if _SideStream.enabled:
                # only the GEMM goes to the second stream; its partials are summed after the join (reduce())
                cls.sums += _SideStream.run(lambda: gemm_tn_grouped(jobs, reduce=False), jobs)
            else:
                gemm_tn_grouped(jobs)

    @classmethod
    def reduce(cls):
        """After ``_SideStream.join()``: queue the fixed-order sums of the groups that ran on the second stream."""
        sums, cls.sums = cls.sums, []
        for part, sp, out in sums:
            M.reduce_partials(part, sp, out=out, accumulate=True)


def group_wgrads(on):
    flush_wgrads()
    _GroupedWgrad.enabled = bool(on)


def flush_wgrads():
    """Launch the queued weight-gradient group, wait for the weight-gradient stream, queue the partial sums."""
    _GroupedWgrad.flush()
    _SideStream.join()
    _GroupedWgrad.reduce()


def linear_wgrad(g2, a2, W=None, splits=None):
    """dW (N, K) fp32 = g2 (M, N)^T a2 (M, K), deterministic split-K; accumulates into W's flat .grad if present."""
    if _mfma_ok(g2, a2) and g2.shape[1] % 8 == 0 and a2.shape[1] % 8 == 0 and g2.shape[0] % 64 == 0:
        gdir = _direct_grad(W) if W is not None else None
        if gdir is not None and _GroupedWgrad.enabled and splits is None and g2.is_contiguous() and a2.is_contiguous():
            _GroupedWgrad.add(g2, a2, gdir.view(-1))
            return None
        if gdir is not None:
            gemm_tn(g2, a2, splits=splits, out=gdir.view(-1), accumulate=True, defer=not _SideStream.enabled)
            return None
        return gemm_tn(g2, a2, splits=splits)
    return _wgrad(g2, a2, W)


class LinearFn(torch.autograd.Function):
    """y = a @ W^T (+ bias, added in the GEMM epilogue) through the MFMA GEMMs, with the deterministic split-K
    weight gradient (and direct accumulation into a flat .grad).  Used for the patch-embed projection and the
    classification head."""

    @staticmethod
    def forward(ctx, a, W, cdt, bias=None):
        with torch.autocast("cuda", enabled=False):
            a2 = a.reshape(-1, a.shape[-1]).to(cdt).contiguous()
            W2 = W.reshape(W.shape[0], -1)
            y = linear_fwd(a2, _shadow(W, cdt).reshape(W2.shape), bias)
        ctx.save_for_backward(a2, W)
        ctx.a_shape, ctx.a_dtype, ctx.cdt = a.shape, a.dtype, cdt
        ctx.need_da = a.requires_grad
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        return y.view(*a.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, g):
        a2, W = ctx.saved_tensors
        cdt = ctx.cdt
        with torch.autocast("cuda", enabled=False):
            g2 = g.reshape(-1, g.shape[-1]).to(cdt).contiguous()
            W2s = _shadow(W, cdt).reshape(W.shape[0], -1)
            da = linear_dgrad(g2, W2s).view(ctx.a_shape).to(ctx.a_dtype) if ctx.need_da else None
            dW = linear_wgrad(g2, a2, W)
            if dW is not None:
                dW = dW.view(W.shape)
            db = None
            if ctx.has_bias:
                bias = ctx.bias_ref
                gd = _direct_grad(bias) if bias is not None else None
                if g2.is_cuda and g2.dtype in (torch.float32, torch.bfloat16):
                    from .glue_ops import column_sum
                    if gd is not None:
                        column_sum(g2, out=gd.view(-1), accumulate=True)     # straight into the flat gradient
                    else:
                        db = column_sum(g2)
                else:
                    db = g2.float().sum(0)
        return da, dW, None, db


def linear_module(mod, x):
    """``mod(x)`` for an ``nn.Linear`` through the build's own GEMMs (``LinearFn``: MFMA kernel + deterministic split-K
    weight gradient, bias in the epilogue) -- what the classification heads and the MAE decoder's embedding / prediction
    layers call instead of ``F.linear``; CPU tensors (checkpoint tooling, tests without a GPU) take the module's own path."""
    if isinstance(mod, nn.Linear) and x.is_cuda:
        return LinearFn.apply(x, mod.weight, _compute_dtype(x), mod.bias)
    return mod(x)


class OutProjAddNormFn(torch.autograd.Function):
    """(g (B, L, d_in) bf16, W_out (d, d_in), residual (B, L, d) fp32, norm weight (d)) -> (normed (B, L, d) bf16,
    residual_out fp32): a mixer's ``out_proj`` (mamba_simple_faster.py:435-444) and the NEXT block's DropPath scale +
    residual add + RMSNorm (models/fastvim.py:168-190) in one GEMM whose epilogue owns whole rows -- bit for bit
    ``gemm_nt`` followed by ``fv_add_norm_fwd``, without the out_proj output ever reaching HBM.  Backward is the two
    adjoints back to back (add + norm, then the data and weight gradients of out_proj)."""

    @staticmethod
    def forward(ctx, g, W_out, residual, norm_w, eps, row_scale, cdt):
        L.require_gpu(g, W_out, residual, norm_w)
        B, Ltok, d_in = g.shape
        d = W_out.shape[0]
        Mrows = B * Ltok
        with torch.autocast("cuda", enabled=False):
            y, res_out, rstd, w32, row_scale, rows_per_scale = _out_proj_add_norm_fwd(g, W_out, residual, norm_w, eps, row_scale, cdt)
        ctx.save_for_backward(g, W_out, res_out, w32, rstd, row_scale)
        ctx.rows_per_scale = rows_per_scale
        ctx.cdt = cdt
        ctx.shape = (B, Ltok, d)
        ctx.w_param = norm_w
        return y.view(B, Ltok, d), res_out.view(B, Ltok, d)

    @staticmethod
    def backward(ctx, dy, dres_out):
        g, W_out, r, w32, rstd, row_scale = ctx.saved_tensors
        B, Ltok, d = ctx.shape
        Mrows, d_in = B * Ltok, g.shape[2]
        cdt, dev = ctx.cdt, g.device
        with torch.autocast("cuda", enabled=False):
            dy = dy.reshape(Mrows, d).contiguous()
            dres_out = dres_out.reshape(Mrows, d).contiguous() if dres_out is not None else None
            dx = torch.empty(Mrows, d, device=dev, dtype=cdt)            # gradient of the out_proj output
            dres_in = torch.empty(Mrows, d, device=dev, dtype=torch.float32)
            lib = L.lib()
            nb = lib.fv_add_norm_blocks(L.i32(Mrows))
            pw = torch.empty(nb, d, device=dev, dtype=torch.float32)
            rc = lib.fv_add_norm_bwd(
                L.ptr(dy), L.i32(L.dtype_code(dy.dtype)), L.ptr(dres_out),
                L.i32(L.dtype_code(dres_out.dtype) if dres_out is not None else 0), L.ptr(r), L.i32(L.FV_F32),
                L.ptr(w32), L.ptr(None), L.ptr(rstd), L.ptr(row_scale), L.i32(ctx.rows_per_scale), L.ptr(dx),
                L.i32(L.dtype_code(cdt)), L.ptr(dres_in), L.i32(L.FV_F32), L.ptr(pw), L.ptr(None), L.i32(Mrows), L.i32(d),
                L.i32(1), L.stream_of(r))
            L.check(rc, "add_norm_bwd")
            gd = _direct_grad(ctx.w_param)
            if gd is not None:
                M.reduce_partials(pw, nb, out=gd.view(-1), accumulate=True)
                dw = None
            else:
                dw = M.reduce_partials(pw, nb).to(ctx.w_param.dtype)
            g2 = g.view(Mrows, d_in)
            dg = linear_dgrad(dx, _shadow(W_out, cdt)).view(B, Ltok, d_in)
            dW_out = _SideStream.run(lambda: linear_wgrad(dx, g2, W_out), dx, g)
        return dg, dW_out, dres_in.view(B, Ltok, d), dw, None, None, None


class _Ctx:
    """Stand-in for an autograd context when one Function runs another's forward / backward as plain code."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


COMBINE_IN_OUT_PROJ = True      # combine (expand + LayerNorm + gate) as the A-tile producer of the out_proj + add + norm launch (A/B switch)
CONV_IN_DGRAD = True            # the conv + pool adjoint as the A-tile producer of the in_proj data gradient + norm adjoint (A/B switch)
XPROJ_TWO_ADDENDS = True        # wide models: the x_proj adjoint's product as a second (bf16) addend of the pooled gradient (A/B switch)


def resolve_combine(pack):
    """Run a deferred combine as its own launch (the chain ends, is cut, or meets a consumer that cannot produce it)."""
    if pack is not None:
        M.combine_fwd(pack["xz"], pack["skip"], pack["yc"], pack["ln_w"], pack["ln_b"], pack["eps"], pack["rows"],
                      pack["cols"], pack["transposed"], out=pack["out"])


def _out_proj_add_norm_fwd(g, W_out, residual, norm_w, eps, row_scale, cdt, W_in=None, pack=None):
    """``W_in`` (2 d_inner, d): also run this block's in_proj as a second phase of the launch; returns xz last.
    ``pack``: g has not been computed yet -- its combine runs inside this launch (``fv_mixer_combine_out_proj_addnorm``)."""
    B, Ltok, d_in = g.shape
    d = W_out.shape[0]
    Mrows = B * Ltok
    g2 = g.view(Mrows, d_in)
    res2 = residual.reshape(Mrows, d).contiguous()
    w32 = norm_w.float().contiguous()
    y = torch.empty(Mrows, d, device=g.device, dtype=cdt)
    res_out = torch.empty(Mrows, d, device=g.device, dtype=torch.float32)
    rstd = torch.empty(Mrows, device=g.device, dtype=torch.float32)
    rows_per_scale = 1
    if row_scale is not None:
        row_scale = row_scale.float().contiguous()
        rows_per_scale = Mrows // row_scale.numel()
    xz = None
    if W_in is not None and W_in.shape[0] % 128 == 0:
        xz = torch.empty(Mrows, W_in.shape[0], device=g.device, dtype=cdt)
    # (the compute-dtype weights are held in locals across the launch: without a flat training state ``_shadow`` returns a
    #  fresh cast, and a temporary inside the argument list is freed -- and its memory handed to the next cast -- before
    #  the kernel is even enqueued)
    W_out_c = _shadow(W_out, cdt)
    W_in_c = _shadow(W_in, cdt) if xz is not None else None
    lib = L.lib()
    if pack is not None:
        if xz is None and W_out_c.stride(1) == 1 and W_out_c.stride(0) % 8 == 0 and W_out_c.data_ptr() % 16 == 0:
            y, res_out, rstd = M.combine_out_proj_addnorm(pack["xz"], pack["skip"], pack["yc"], pack["ln_w"], pack["ln_b"],
                                                          pack["eps"], pack["rows"], pack["cols"], pack["transposed"],
                                                          pack["out"], W_out_c, res2, w32, row_scale, rows_per_scale, eps)
            return y, res_out, rstd, w32, row_scale, rows_per_scale
        resolve_combine(pack)
    rc = lib.fv_gemm_bf16_addnorm2(
        L.ptr(g2), L.ptr(W_out_c), L.ptr(res2), L.ptr(w32), L.ptr(row_scale), L.i32(rows_per_scale),
        L.ptr(y), L.ptr(res_out), L.ptr(rstd), L.i32(Mrows), L.i32(d), L.i32(d_in), ctypes.c_long(g2.stride(0)),
        ctypes.c_long(d_in), ctypes.c_float(eps), L.ptr(W_in_c), L.ptr(xz),
        L.i32(W_in.shape[0] if xz is not None else 0), ctypes.c_long(d), L.stream_of(g2))
    L.check(rc, "gemm_bf16_addnorm")
    if W_in is not None:
        return y, res_out, rstd, w32, row_scale, rows_per_scale, xz
    return y, res_out, rstd, w32, row_scale, rows_per_scale


class ChainedBlockFn(torch.autograd.Function):
    """One block of a chained run (``VisionMamba._run_layers_chained``), d_model = 192: the previous mixer's ``out_proj``
    + this block's DropPath scale / residual add / RMSNorm (``fv_gemm_bf16_addnorm``), then this block's mixer up to its
    gated activations.  Inputs: the previous block's gated activations and ``out_proj`` weight, the residual stream, the
    norm weight, then ``FastVimMixerFn``'s arguments after ``hidden`` (with ``W_out`` / ``b_out`` None).  Returns
    (gated activations, residual_out).  Backward runs the mixer's adjoint, with the in_proj data gradient produced
    INSIDE the norm's adjoint (``fv_gemm_bf16_dgrad_addnorm_bwd2``: it is needed by nothing else) and the previous
    ``out_proj``'s data gradient as a second GEMM phase of the same launch (from the d x tile still in LDS), then that
    ``out_proj``'s weight gradient -- the values of ``OutProjAddNormFn`` + ``FastVimMixerFn`` back to back; only the norm
    weight's gradient is summed in a different (fixed) order."""

    # this block's in_proj as a second GEMM phase of the forward launch (fv_gemm_bf16_addnorm2): bit-identical and 4 us
    # faster stand-alone (32.3 vs 36.3 us), but 0.07 ms SLOWER per FastVim-T step (5.94 vs 5.87 ms, same box): the
    # 38 MB of xz stores at the end of a 392-workgroup launch cost more than the launch they save.  Off.
    in_proj_in_launch = False
    # side channel of the chained run (single-threaded, set and consumed around one ``apply``): ``pending_in`` = the
    # deferred combine of g_prev (its buffers are g_prev's storage), ``pending_out`` = this block's own deferred combine
    pending_in = None
    pending_out = None
    defer = False

    @staticmethod
    def forward(ctx, g_prev, W_out_prev, residual, norm_w, eps, row_scale, *mixer_args):
        L.require_gpu(g_prev, W_out_prev, residual, norm_w)
        B, Ltok, _ = g_prev.shape
        d = W_out_prev.shape[0]
        cdt = mixer_args[26]            # FastVimMixerFn.forward(ctx, hidden, <26 arguments>, cdt, fv, tpp, valid)
        W_in, b_in = mixer_args[0], mixer_args[1]
        pack, ChainedBlockFn.pending_in = ChainedBlockFn.pending_in, None
        assert pack is None or pack["out"][0].data_ptr() == g_prev.data_ptr()
        with torch.autocast("cuda", enabled=False):
            if pack is not None:
                y, res_out, rstd, w32, rs, rps = _out_proj_add_norm_fwd(g_prev, W_out_prev, residual, norm_w, eps, row_scale,
                                                                      cdt, pack=pack)
                xz = None
            elif ChainedBlockFn.in_proj_in_launch and b_in is None:
                y, res_out, rstd, w32, rs, rps, xz = _out_proj_add_norm_fwd(g_prev, W_out_prev, residual, norm_w, eps,
                                                                          row_scale, cdt, W_in=W_in)
            else:
                y, res_out, rstd, w32, rs, rps = _out_proj_add_norm_fwd(g_prev, W_out_prev, residual, norm_w, eps, row_scale, cdt)
                xz = None
        fctx = _Ctx()
        fctx.precomputed_xz = xz
        fctx.defer_combine = ChainedBlockFn.defer
        g = FastVimMixerFn.forward(fctx, y.view(B, Ltok, d), *mixer_args)
        mixer_saved = fctx.__dict__.pop("saved_tensors")
        fctx.__dict__.pop("precomputed_xz", None)
        fctx.__dict__.pop("defer_combine", None)
        ChainedBlockFn.pending_out = fctx.__dict__.pop("pending_combine", None)
        ctx.save_for_backward(g_prev, W_out_prev, res_out, w32, rstd, rs, *mixer_saved)
        ctx.mixer_attrs = fctx.__dict__
        ctx.rows_per_scale = rps
        ctx.cdt = cdt
        ctx.shape = (B, Ltok, d)
        ctx.w_param = norm_w
        ctx.W_in = mixer_args[0]
        ctx.n_mixer_args = len(mixer_args)
        return g, res_out.view(B, Ltok, d)

    @staticmethod
    def backward(ctx, dg, dres_out):
        g_prev, W_out_prev, r, w32, rstd, row_scale = ctx.saved_tensors[:6]
        B, Ltok, d = ctx.shape
        Mrows, cdt, dev = B * Ltok, ctx.cdt, r.device
        fctx = _Ctx()
        fctx.__dict__.update(ctx.mixer_attrs)
        fctx.saved_tensors = ctx.saved_tensors[6:]
        out = {}

        def fused_in_dgrad(dxz2):
            lib = L.lib()
            nb = lib.fv_gemm_bf16_dgrad_addnorm_blocks(L.i32(Mrows))
            dx = torch.empty(Mrows, d, device=dev, dtype=cdt)
            dres_in = torch.empty(Mrows, d, device=dev, dtype=torch.float32)
            pw = torch.empty(nb, d, device=dev, dtype=torch.float32)
            gg = dres_out.reshape(Mrows, d).contiguous() if dres_out is not None else None
            # second phase of the same launch: the previous block's out_proj data gradient, from the d x tile in LDS
            d_prev = g_prev.shape[2]
            W2 = _shadow(W_out_prev, cdt) if d_prev % 128 == 0 else None
            W_in_c = _shadow(ctx.W_in, cdt)                  # (held across the launch, see _out_proj_add_norm_fwd)
            dg_prev = torch.empty(Mrows, d_prev, device=dev, dtype=cdt) if W2 is not None else None
            rc = lib.fv_gemm_bf16_dgrad_addnorm_bwd2(
                L.ptr(dxz2), L.ptr(W_in_c), L.ptr(gg), L.ptr(r), L.ptr(rstd), L.ptr(w32), L.ptr(row_scale),
                L.i32(ctx.rows_per_scale), L.ptr(dx), L.ptr(dres_in), L.ptr(pw), L.i32(Mrows), L.i32(d),
                L.i32(dxz2.shape[1]), ctypes.c_long(dxz2.stride(0)), ctypes.c_long(d), L.ptr(W2), L.ptr(dg_prev),
                L.i32(d_prev if W2 is not None else 0), ctypes.c_long(d_prev), L.stream_of(dxz2))
            L.check(rc, "gemm_bf16_dgrad_addnorm_bwd")
            out.update(dx=dx, dres_in=dres_in, pw=pw, nb=nb, dg_prev=dg_prev)

        def fused_conv_dgrad(xz, d_o, dxc, dxc2, cw2, cb, cwb2, cb_b, D, D_b, dxz, rows, cols, transposed, scaling, conv_grad_out):
            """The conv + pool adjoint AND the launch above in one (fv_mixer_conv_pool_bwd_dgrad); returns the conv
            parameter-gradient sums (None when accumulated into ``conv_grad_out``)."""
            gg = dres_out.reshape(Mrows, d).contiguous() if dres_out is not None else None
            d_prev = g_prev.shape[2]
            W2 = _shadow(W_out_prev, cdt) if d_prev % 128 == 0 else None
            W_in_t = _shadow_t(ctx.W_in, cdt)                # (held across the launch, see _out_proj_add_norm_fwd)
            p2, dx, dres_in, pw, nb, dg_prev = M.conv_pool_bwd_dgrad(
                xz, d_o, dxc, dxc2, cw2, cb, cwb2, cb_b, D, D_b, dxz, rows, cols, transposed, scaling, W_in_t, gg, r, rstd,
                w32, row_scale, ctx.rows_per_scale, W2=W2, conv_grad_out=conv_grad_out)
            out.update(dx=dx, dres_in=dres_in, pw=pw, nb=nb, dg_prev=dg_prev)
            return p2

        fctx.fused_in_dgrad = fused_in_dgrad
        if (CONV_IN_DGRAD and cdt == torch.bfloat16 and d == 192 and ctx.W_in.shape[0] == 768 and (row_scale is None or row_scale.dtype == torch.float32)):
            fctx.fused_conv_dgrad = fused_conv_dgrad
        grads = FastVimMixerFn.backward(fctx, dg)
        with torch.autocast("cuda", enabled=False):
            gd = _direct_grad(ctx.w_param)
            if gd is not None:
                M.reduce_partials(out["pw"], out["nb"], out=gd.view(-1), accumulate=True)
                dw = None
            else:
                dw = M.reduce_partials(out["pw"], out["nb"]).to(ctx.w_param.dtype)
            dx = out["dx"]
            d_in = g_prev.shape[2]
            g2 = g_prev.view(Mrows, d_in)
            dg_prev = out["dg_prev"] if out["dg_prev"] is not None else linear_dgrad(dx, _shadow(W_out_prev, cdt))
            dg_prev = dg_prev.view(B, Ltok, d_in)
            dW_out = _SideStream.run(lambda: linear_wgrad(dx, g2, W_out_prev), dx, g_prev)
        return (dg_prev, dW_out, out["dres_in"].view(B, Ltok, d), dw, None, None) + tuple(grads[1:1 + ctx.n_mixer_args])


def out_proj_add_norm_ok(g, W_out, residual, norm_w, cdt):
    """The fused kernel owns whole 192-wide rows and stages through LDS-DMA (K % 64 == 0)."""
    return (g.is_cuda and cdt == torch.bfloat16 and g.dtype == torch.bfloat16 and g.is_contiguous()
            and W_out.shape[0] == 192 and W_out.shape[1] % 64 == 0 and residual is not None
            and residual.dtype == torch.float32 and norm_w.dtype == torch.float32)


def _compute_dtype(t):
    """The dtype the kernels compute and store activations in: the autocast dtype under torch.autocast (reference:
    mamba_simple_faster.py:312-318), else the input dtype -- except fp16, which is UPCAST: the module path (patch embed,
    blocks, mixers) is built for bf16 and fp32, so the reference's other mixed precision, ``--precision 16-mixed``
    (imagenet_classification/train.py:17), and a ``model.half()`` run in fp32 with fp16 at the model's boundary
    (``half_io``): same function, fp32 (not fp16) rounding inside, loss scaling passes through untouched because no
    intermediate is ever stored in fp16.  The op-level functions (``selective_scan_fn``, ``causal_conv1d_fn``, the compressed
    scan) take fp16 tensors natively, like the reference's kernels (selective_scan.cpp:328-332)."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype('cuda')
        return torch.float32 if dt == torch.float16 else dt
    return torch.float32 if t.dtype == torch.float16 else t.dtype


def half_io(t):
    """Is this call in the fp16 regime (fp16 autocast, or fp16 activations / a ``.half()`` model)?  Module outputs are
    then cast to fp16, what the reference returns there."""
    if torch.is_autocast_enabled():
        return torch.get_autocast_dtype('cuda') == torch.float16
    return t.dtype == torch.float16


def mixer_apply(fn, hidden, *args):
    """``fn.apply`` for the mixer Functions with fp16 PARAMETERS (a ``model.half()``) upcast first: the kernels read fp32
    masters through raw pointers.  The casts are autograd nodes, so a fp16 parameter still gets its (fp16) gradient."""
    if any(torch.is_tensor(a) and a.dtype == torch.float16 for a in args):
        args = tuple(a.float() if torch.is_tensor(a) and a.dtype == torch.float16 else a for a in args)
    out = fn.apply(hidden, *args)
    # fp16 activations in, fp16 out.  (Under fp16 AUTOCAST with fp32 activations -- what the blocks of this build's models
    # hand over -- the result stays fp32: the norm kernels between blocks take fp32 / bf16 rows, and the model casts once
    # at its boundary.)
    return out.to(torch.float16) if hidden.dtype == torch.float16 else out


class FastVimMixerFn(torch.autograd.Function):
    """hidden (B, L, d) -> out (B, L, d).  Geometry: the mixer's (rows, cols) pooling grid;
    ``transposed`` = memory tokens are laid out as the (cols, rows) grid (odd layers)."""

    @staticmethod
    def forward(ctx, hidden, W_in, b_in, cw, cb, cw_b, cb_b, Wx, Wx_b, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log,
                D, D_b, ln_w, ln_b, W_out, b_out, rows, cols, transposed, pool_max, scaling, ln_eps, cdt, fv, tpp=1,
                valid=None):
        """``valid``: number of pooled positions that are real; the rest are zero-input padding at the END of the sequence
        (the un-pooled Vim mixer pads 197 tokens to 25 rows of 8).  Their pooled conv output is zeroed, which makes them
        inert in both scans (u = 0 => B = C = 0; the state they see first is zero), and nothing else has to know."""
        L.require_gpu(hidden)
        B, Ltok, d = hidden.shape
        if Ltok != rows * cols * tpp:
            raise RuntimeError(f"Mamba: sequence length {Ltok} != token grid {rows}x{cols}x{tpp}")
        d_in = W_in.shape[0] // 2
        with torch.autocast("cuda", enabled=False):
            h_c = hidden.to(cdt).contiguous()
            W_in_c = _shadow(W_in, cdt)
            W_out_c = _shadow(W_out, cdt) if W_out is not None else None
            xz = getattr(ctx, "precomputed_xz", None)          # ChainedBlockFn: in_proj ran inside the previous launch
            if xz is None:
                xz = linear_fwd(h_c.view(B * Ltok, d), W_in_c, b_in)
            xz = xz.view(B, Ltok, 2 * d_in)                                                # (B, L, 2 d_in)
            cw2, cwb2 = cw.reshape(d_in, -1), cw_b.reshape(d_in, -1)
            amax = None
            if fv is not None and "Wx2" in fv:          # x_proj / x_proj_b adjacent in the flat buffers
                Wx2 = fv["Wx2"]
                if fv["Wx2_shadow"].dtype == cdt:
                    _shadow(Wx, cdt), _shadow(Wx_b, cdt)       # version check of the two halves of Wx2_shadow
                    Wx2_c = fv["Wx2_shadow"]
                else:
                    Wx2_c = Wx2.to(cdt)
            else:
                Wx2 = torch.stack([Wx, Wx_b])                                           # (2, R+2N, d_in) fp32
                Wx2_c = Wx2.to(cdt)
            if pool_max:
                xc, skip, amax = M.conv_pool_fwd(xz, cw2, cb, cwb2, cb_b, rows, cols, transposed, pool_max, scaling, tpp,
                                                 D=D, D_b=D_b)
            else:
                xc, skip = M.conv_pool_fwd(xz, cw2, cb, cwb2, cb_b, rows, cols, transposed, pool_max, scaling, tpp,
                                           D=D, D_b=D_b)
            if valid is not None and valid < rows * tpp:
                xc[:, :, valid:].zero_()
            fused = M.xproj_scan_fwd(xc, Wx2_c, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log)      # short pooled lengths, bf16
            if fused is not None:
                x_dbl, yc = fused
            else:
                x_dbl = M.xproj_fwd(xc, Wx2_c)                                           # (2, B*Lc, R+2N)
                # long pooled lengths: when a backward pass will follow, the scan leaves the state entering every
                # 16-step chunk behind and the backward kernel does not sweep forward again
                nig = getattr(ctx, "needs_input_grad", None)
                yc, ctx.scan_ckpt = M.scan_fwd(xc, x_dbl, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log,
                                               want_ckpt=nig is None or any(nig))
            if (getattr(ctx, "defer_combine", False) and W_out is None and not pool_max and valid is None
                    and M.combine_out_proj_addnorm_ok(xz, rows, cols, tpp, d)):
                # the caller (ChainedBlockFn of the NEXT block) gates the activations inside its out_proj + add + norm
                # launch: the buffers exist (they are this node's output / saved tensors), their contents come later
                g, mean, rstd = M.combine_buffers(xz, ln_w)
                ctx.pending_combine = dict(xz=xz, skip=skip, yc=yc, ln_w=ln_w, ln_b=ln_b, eps=ln_eps, rows=rows, cols=cols,
                                           transposed=transposed, out=(g, mean, rstd))
            else:
                g, mean, rstd = M.combine_fwd(xz, skip, yc, ln_w, ln_b, ln_eps, rows, cols, transposed, tpp=tpp)
            # W_out None: out_proj is the caller's (fused with the next block's add + norm, OutProjAddNormFn); the
            # gated activations g (B, L, d_in) are returned and their gradient comes back as ``dout``
            out = g if W_out is None else linear_fwd(g.view(B * Ltok, d_in), W_out_c, b_out).view(B, Ltok, d)
        ctx.save_for_backward(h_c, W_in, cw, cb, cw_b, cb_b, Wx2, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log, D, D_b,
                              ln_w, ln_b, W_out, xz, xc, x_dbl, g, skip, yc, mean, rstd, amax)
        ctx.geo = (rows, cols, transposed, pool_max, scaling, tpp)
        ctx.has_bias = (b_in is not None, b_out is not None)
        ctx.cdt = cdt
        ctx.in_dtype = hidden.dtype
        ctx.fv = fv
        return out

    @staticmethod
    def backward(ctx, dout):
        (h_c, W_in, cw, cb, cw_b, cb_b, Wx2, Wdt, bdt, Wdt_b, bdt_b, A_log, A_b_log, D, D_b, ln_w, ln_b, W_out,
         xz, xc, x_dbl, g, skip, yc, mean, rstd, amax) = ctx.saved_tensors
        rows, cols, transposed, pool_max, scaling, tpp = ctx.geo
        cdt = ctx.cdt
        B, Ltok, d = h_c.shape
        d_in = W_in.shape[0] // 2
        fv = ctx.fv or {}
        with torch.autocast("cuda", enabled=False):
            dout = dout.to(cdt).contiguous()
            if W_out is None:
                dg, dW_out, db_out = dout.view(B * Ltok, d_in), None, None
            else:
                do2 = dout.view(B * Ltok, d)
                dg = linear_dgrad(do2, _shadow(W_out, cdt))                              # (B*L, d_in)
                dW_out = _SideStream.run(lambda: linear_wgrad(do2, g.view(B * Ltok, d_in), W_out), do2, g)
                db_out = do2.float().sum(0) if ctx.has_bias[1] else None
            cw2, cwb2 = cw.reshape(d_in, -1), cw_b.reshape(d_in, -1)
            dxz = torch.empty_like(xz)
            d_o, dyc, p1 = M.combine_bwd(dg, xz, skip, yc, ln_w, ln_b, mean, rstd, dxz, rows, cols, transposed,
                                         grad_out=fv.get("ln_grad") if ln_w is not None else None, tpp=tpp)
            W_ = x_dbl.shape[-1]
            fused_xproj = W_ in M.XPROJ_WIDTHS
            Mrows = B * rows * tpp
            grouped_x = (fused_xproj and _GroupedWgrad.enabled and "Wx2_grad" in fv and xc.dtype == torch.bfloat16
                         and Mrows % 64 == 0 and d_in % 8 == 0)
            dxc2 = None
            if (grouped_x and XPROJ_IN_SCAN and amax is None and getattr(ctx, "scan_ckpt", None) is None
                    and M.scan_bwd_xproj_ok(xc, Wdt, pool_max, rows, cols, tpp)):
                # FastVim-T: the x_proj adjoint's data half runs inside the scan backward (one launch less per block); the
                # pooled gradient arrives as two addends, the bf16 d x_dbl rows of the weight gradient are made by one
                # launch for all blocks right before the grouped weight-gradient GEMMs
                wsh = fv.get("Wx2_shadow")
                dxc, dxc2, dx_dbl, ps = M.scan_bwd_xproj(xc, x_dbl, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log, dyc,
                                                         Wx2[0], Wx2[1], grad_out=fv.get("scan_grad"),
                                                         Wx2_bf16=wsh if wsh is not None and wsh.dtype == torch.bfloat16 else None)
                dxb = torch.empty(2, Mrows, (W_ + 7) // 8 * 8, device=xc.device, dtype=torch.bfloat16)
                _GroupedWgrad.rows_jobs.append((dx_dbl, dxb))
                xc2 = xc.view(2, Mrows, d_in)
                for k_ in range(2):
                    _GroupedWgrad.add(dxb[k_][:, :W_], xc2[k_], fv["Wx2_grad"][k_].reshape(-1))
                dWx2 = (None, None)
            else:
                dxc, dx_dbl, ps = M.scan_bwd(xc, x_dbl, Wdt, bdt, A_log, Wdt_b, bdt_b, A_b_log, dyc,
                                             grad_out=fv.get("scan_grad"), keep_chunks=fused_xproj,
                                             ckpt=getattr(ctx, "scan_ckpt", None))
            # x_proj adjoint (selective_scan_interface.py:726-734), both directions
            if dxc2 is not None:
                pass
            elif grouped_x:
                # the weight gradient dx_dbl^T xc joins the grouped launch at the end of backward (bf16 dx_dbl, as in
                # the reference's autocast backward); the kernel only adds dx_dbl @ Wx to dxc
                # wide models (flat training state): the product runs on the bf16 matrix cores from the transposed shadow
                # weight, and where the conv + pool adjoint takes a second addend (14- / 16-column grids) it is written
                # as its own bf16 tensor instead of a read-modify-write of the fp32 d xc
                wt = _wx2_shadow_t(fv)
                if (wt is not None and amax is None and XPROJ_TWO_ADDENDS and M.xproj_bwd3_ok(Mrows, d_in, W_, xc.dtype)
                        and M.conv_pool_bwd2_ok(rows, cols, tpp, d_in, pool_max)):
                    dxc2 = torch.empty(dxc.shape, device=dxc.device, dtype=xc.dtype)
                dxb = M.xproj_bwd(dx_dbl, xc, Wx2[0], Wx2[1], dxc, dw=False, Wx2_t=wt, dxc2=dxc2)
                xc2 = xc.view(2, Mrows, d_in)
                for k_ in range(2):
                    _GroupedWgrad.add(dxb[k_][:, :W_], xc2[k_], fv["Wx2_grad"][k_].reshape(-1))
                dWx2 = (None, None)
            elif fused_xproj:
                dWx2 = M.xproj_bwd(dx_dbl, xc, Wx2[0], Wx2[1], dxc, grad_out=fv.get("Wx2_grad"))
                if dWx2 is None:
                    dWx2 = (None, None)
            else:
                # widths the x_proj adjoint kernel is not built for: the fp32-MFMA GEMM, both directions as its batch
                xc2 = xc.view(2, B * rows * tpp, d_in)
                dWx2 = gemm_any_btn(dx_dbl, xc2)                                          # (2, R+2N, d_in) fp32
                if "Wx2_grad" in fv:
                    fv["Wx2_grad"].add_(dWx2)
                    dWx2 = (None, None)
                dxc = dxc.view(2, B * rows * tpp, d_in) + gemm_any_bnn(dx_dbl, Wx2.float(), out_dtype=torch.float32)   # + dx_dbl @ Wx
            conv_grad = fv.get("conv_grad") if cb is not None and cb_b is not None else None
            fused_cd = getattr(ctx, "fused_conv_dgrad", None)
            dxz2 = dxz.view(B * Ltok, 2 * d_in)
            if (fused_cd is not None and amax is None and dxc.dtype == torch.float32 and dxc.is_contiguous()
                    and M.conv_pool_bwd_dgrad_ok(xz, rows, cols, tpp, d, pool_max)):
                # ChainedBlockFn, FastVim-T: the conv + pool adjoint is the A-tile producer of the in_proj data gradient,
                # whose epilogue is the norm's adjoint (one launch instead of two; the x half of dxz is still written --
                # the weight gradient below reads it)
                p2 = fused_cd(xz, d_o, dxc, dxc2, cw2, cb, cwb2, cb_b, D, D_b, dxz, rows, cols, transposed, scaling, conv_grad)
                fused_dgrad = "done"
            else:
                p2 = M.conv_pool_bwd(xz, d_o, dxc, cw2, cb, cwb2, cb_b, D, D_b, dxz, rows, cols, transposed,
                                     pool_max, scaling, grad_out=conv_grad, tpp=tpp, amax=amax, dxc2=dxc2)
                fused_dgrad = getattr(ctx, "fused_in_dgrad", None)
            if fused_dgrad == "done":
                dhidden = None
            elif fused_dgrad is not None:
                fused_dgrad(dxz2)          # ChainedBlockFn: the data gradient goes straight into the norm's adjoint
                dhidden = None
            else:
                dhidden = linear_dgrad(dxz2, _shadow(W_in, cdt)).view(B, Ltok, d).to(ctx.in_dtype)
            dW_in = _SideStream.run(lambda: linear_wgrad(dxz2, h_c.view(B * Ltok, d), W_in), dxz2, h_c)
            db_in = dxz2.float().sum(0) if ctx.has_bias[0] else None
        has_ln = ln_w is not None
        n4 = 4 * d_in
        N_, R_ = A_log.shape[1], Wdt.shape[1]
        if p2 is None:
            g_cw = g_cb = g_cwb = g_cbb = g_D = g_Db = None
        else:
            g_cw, g_cwb = p2[0:n4].view(cw.shape), p2[n4:2 * n4].view(cw_b.shape)
            g_cb = p2[2 * n4:2 * n4 + d_in] if cb is not None else None
            g_cbb = p2[2 * n4 + d_in:2 * n4 + 2 * d_in] if cb_b is not None else None
            g_D, g_Db = p2[2 * n4 + 2 * d_in:2 * n4 + 3 * d_in], p2[2 * n4 + 3 * d_in:2 * n4 + 4 * d_in]
        if ps is None:
            g_A = g_Wdt = g_bdt = (None, None)
        else:
            g_A = [ps[k, :d_in * N_].view(d_in, N_) for k in range(2)]
            g_Wdt = [ps[k, d_in * N_:d_in * (N_ + R_)].view(d_in, R_) for k in range(2)]
            g_bdt = [ps[k, d_in * (N_ + R_):] for k in range(2)]
        g_lw = p1[0] if (has_ln and p1 is not None) else None
        g_lb = p1[1] if (has_ln and p1 is not None) else None
        return (dhidden, dW_in, db_in, g_cw, g_cb, g_cwb, g_cbb,
                dWx2[0], dWx2[1], g_Wdt[0], g_bdt[0], g_Wdt[1], g_bdt[1], g_A[0], g_A[1],
                g_D, g_Db, g_lw, g_lb, dW_out, db_out, None, None, None, None, None, None, None, None, None, None)


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=False, layer_idx=None, device=None, dtype=None, init_layer_scale=None,
                 scanpath_type="rowwise", token_size=None, use_norm_after_ssm=True,
                 use_our_selective_scan=False, collapse_method="mean", scaling_factor=1):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path            # accepted for config compatibility; one fused path here
        self.layer_idx = layer_idx
        self.use_our_selective_scan = use_our_selective_scan
        self.num_of_rows, self.num_of_col = token_size[0], token_size[1]
        self.scanpath_type = scanpath_type
        self.scaling_factor = scaling_factor
        if collapse_method not in ("mean", "max"):
            raise NotImplementedError(collapse_method)
        self.collapse_method = collapse_method
        self.init_layer_scale = init_layer_scale
        if init_layer_scale is not None:
            self.gamma = nn.Parameter(init_layer_scale * torch.ones(d_model), requires_grad=True)

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.use_norm_after_ssm = use_norm_after_ssm
        if use_norm_after_ssm:
            self.layernorm = nn.LayerNorm(self.d_inner, **factory_kwargs)

        def conv():
            return nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                             padding=d_conv - 1, **factory_kwargs)

        self.conv1d = conv()
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)

        # dt projection init preserving variance; bias = softplus^-1(dt), dt ~ logU[dt_min, dt_max]
        # (mamba_simple_faster.py:110-130)
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True

        def s4d_real():
            A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1)
            p = nn.Parameter(torch.log(A).contiguous())       # fp32
            p._no_weight_decay = True
            return p

        self.A_log = s4d_real()
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.A_b_log = s4d_real()
        self.conv1d_b = conv()
        self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D_b._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)
        self.pre_x_shape = (-1, self.d_inner, self.num_of_rows, self.num_of_col)

    def mixer_fn_args(self, cdt, transposed_grid=False, defer_out_proj=False):
        """``FastVimMixerFn``'s arguments after ``hidden`` for this module."""
        ln_w = self.layernorm.weight if self.use_norm_after_ssm else None
        ln_b = self.layernorm.bias if self.use_norm_after_ssm else None
        ln_eps = self.layernorm.eps if self.use_norm_after_ssm else 0.0
        return (self.in_proj.weight, self.in_proj.bias,
                self.conv1d.weight, self.conv1d.bias, self.conv1d_b.weight, self.conv1d_b.bias,
                self.x_proj.weight, self.x_proj_b.weight,
                self.dt_proj.weight, self.dt_proj.bias, self.dt_proj_b.weight, self.dt_proj_b.bias,
                self.A_log, self.A_b_log, self.D, self.D_b, ln_w, ln_b,
                None if defer_out_proj else self.out_proj.weight, None if defer_out_proj else self.out_proj.bias,
                self.num_of_rows, self.num_of_col, bool(transposed_grid), self.collapse_method == "max",
                float(self.scaling_factor), float(ln_eps), cdt, self.__dict__.get("_fv"))

    def forward(self, hidden_states, inference_params=None, transposed_grid=False, defer_out_proj=False):
        """hidden_states: (B, L, D) -> (B, L, D).  ``defer_out_proj``: return the gated activations (B, L, d_inner)
        instead -- ``out_proj`` is then applied by the caller (``OutProjAddNormFn``: fused with the next block's add + norm).

        ``transposed_grid=False``: tokens are in this mixer's own (rows, cols) sequence order, exactly
        the reference contract.  ``transposed_grid=True``: tokens are in the *transposed* (cols, rows)
        order -- what ``Block`` holds in memory on odd layers -- and the output comes back in that same
        order; equivalent to transpose -> mixer -> transpose (models/fastvim.py:192-210) with no copies."""
        if inference_params is not None:
            raise NotImplementedError("FastVim mixers have no inference cache (reference: no step())")
        if self.d_conv != 4 or self.d_state != 16:
            raise RuntimeError("fastvim_amd kernels are built for d_conv=4, d_state=16 (the FastVim configs)")
        cdt = _compute_dtype(hidden_states)
        ln_w = self.layernorm.weight if self.use_norm_after_ssm else None
        ln_b = self.layernorm.bias if self.use_norm_after_ssm else None
        ln_eps = self.layernorm.eps if self.use_norm_after_ssm else 0.0
        out = mixer_apply(
            FastVimMixerFn, hidden_states, self.in_proj.weight, self.in_proj.bias,
            self.conv1d.weight, self.conv1d.bias, self.conv1d_b.weight, self.conv1d_b.bias,
            self.x_proj.weight, self.x_proj_b.weight,
            self.dt_proj.weight, self.dt_proj.bias, self.dt_proj_b.weight, self.dt_proj_b.bias,
            self.A_log, self.A_b_log, self.D, self.D_b, ln_w, ln_b,
            None if defer_out_proj else self.out_proj.weight, None if defer_out_proj else self.out_proj.bias,
            self.num_of_rows, self.num_of_col, bool(transposed_grid), self.collapse_method == "max",
            float(self.scaling_factor), float(ln_eps), cdt, self.__dict__.get("_fv"))
        if defer_out_proj:
            if self.init_layer_scale is not None or self.out_proj.bias is not None:
                raise RuntimeError("defer_out_proj: layer scale / out_proj bias are applied after out_proj")
            return out
        if self.init_layer_scale is not None:
            out = out * self.gamma
        return out
