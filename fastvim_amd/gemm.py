"""Host wrappers of the bf16 MFMA GEMM (csrc/gemm_mfma.hip, C ABI fv_gemm_bf16)."""
import ctypes

import torch

from . import _lib as L
from .mixer_ops import reduce_partials


def _call(A, B, C, bias, M, N, K, lda, ldb, ldc, a_ks, b_ks, splits):
    rc = L.lib().fv_gemm_bf16(L.ptr(A), L.ptr(B), L.ptr(C), L.ptr(bias), L.i32(M), L.i32(N), L.i32(K),
                              ctypes.c_long(lda), ctypes.c_long(ldb), ctypes.c_long(ldc), L.i32(a_ks), L.i32(b_ks),
                              L.i32(C.dtype == torch.float32), L.i32(splits), L.stream_of(A))
    L.check(rc, "gemm_bf16")


def gemm_f32(A, B, C, bias, M, N, K, lda, ldb, ldc, a_ks, b_ks, batch=1, sA=0, sB=0, sC=0):
    """The fp32-MFMA GEMM (csrc/gemm_f32.hip, C ABI fv_gemm_f32): fp32 or bf16 operands of any shape / alignment."""
    L.require_gpu(A, B, C)
    rc = L.lib().fv_gemm_f32(L.ptr(A), L.i32(L.dtype_code(A.dtype)), L.ptr(B), L.i32(L.dtype_code(B.dtype)), L.ptr(C),
                             L.i32(L.dtype_code(C.dtype)), L.ptr(bias), L.i32(M), L.i32(N), L.i32(K), ctypes.c_long(lda),
                             ctypes.c_long(ldb), ctypes.c_long(ldc), L.i32(a_ks), L.i32(b_ks), L.i32(batch),
                             ctypes.c_long(sA), ctypes.c_long(sB), ctypes.c_long(sC), L.stream_of(A))
    L.check(rc, "gemm_f32")


def _rows(t):
    """(tensor with unit inner stride, row stride): what the generic kernel needs of a 2-D operand."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t, t.stride(0)


def gemm_any_nt(a, w, bias=None, out_dtype=None):
    """a (M, K) @ w (N, K)^T (+ bias) through the fp32-MFMA kernel: fp32 operands, or bf16 ones the tuned kernels do not take."""
    (a, lda), (w, ldw) = _rows(a), _rows(w)
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, device=a.device, dtype=out_dtype or a.dtype)
    gemm_f32(a, w, c, None if bias is None else bias.float().contiguous(), M, N, K, lda, ldw, N, 0, 0)
    return c


def gemm_any_nn(a, b, out_dtype=None):
    """a (M, K) @ b (K, N), b row-major as stored."""
    (a, lda), (b, ldb) = _rows(a), _rows(b)
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, device=a.device, dtype=out_dtype or a.dtype)
    gemm_f32(a, b, c, None, M, N, K, lda, ldb, N, 0, 1)
    return c


def gemm_any_tn(x, y, splits=1):
    """x (Kd, M)^T @ y (Kd, N) -> (splits, M, N) fp32 partials over `splits` equal slices of Kd (the last one takes the
    remainder rows when Kd does not divide: then ``splits`` is cut to 1)."""
    (x, ldx), (y, ldy) = _rows(x), _rows(y)
    Kd, M = x.shape
    N = y.shape[1]
    if splits < 1 or Kd % splits:
        splits = 1
    ks = Kd // splits
    part = torch.empty(splits, M, N, device=x.device, dtype=torch.float32)
    gemm_f32(x, y, part, None, M, N, ks, ldx, ldy, N, 1, 1, batch=splits, sA=ks * ldx, sB=ks * ldy, sC=M * N)
    return part


def gemm_any_bnt(a, w, out_dtype=None):
    """Batched a (G, M, K) @ w (G, N, K)^T -> (G, M, N) in one launch (x_proj of both scan directions)."""
    a, w = a.contiguous(), w.contiguous()
    G, M, K = a.shape
    N = w.shape[1]
    c = torch.empty(G, M, N, device=a.device, dtype=out_dtype or a.dtype)
    gemm_f32(a, w, c, None, M, N, K, K, K, N, 0, 0, batch=G, sA=M * K, sB=N * K, sC=M * N)
    return c


def gemm_any_bnn(a, b, out_dtype=None):
    """Batched a (G, M, K) @ b (G, K, N) -> (G, M, N), b row-major as stored."""
    a, b = a.contiguous(), b.contiguous()
    G, M, K = a.shape
    N = b.shape[2]
    c = torch.empty(G, M, N, device=a.device, dtype=out_dtype or a.dtype)
    gemm_f32(a, b, c, None, M, N, K, K, N, N, 0, 1, batch=G, sA=M * K, sB=K * N, sC=M * N)
    return c


def gemm_any_btn(x, y):
    """Batched x (G, Kd, M)^T @ y (G, Kd, N) -> (G, M, N) fp32."""
    x, y = x.contiguous(), y.contiguous()
    G, Kd, M = x.shape
    N = y.shape[2]
    c = torch.empty(G, M, N, device=x.device, dtype=torch.float32)
    gemm_f32(x, y, c, None, M, N, Kd, M, N, N, 1, 1, batch=G, sA=Kd * M, sB=Kd * N, sC=M * N)
    return c


def gemm_nt(a, w, bias=None, out_dtype=torch.bfloat16):
    """a (M, K) @ w (N, K)^T -> (M, N): F.linear(a, w, bias) with bf16 operands."""
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty(M, N, device=a.device, dtype=out_dtype)
    _call(a, w, c, bias, M, N, K, a.stride(0), w.stride(0), N, 0, 0, 1)
    return c


def gemm_nn(a, b, out_dtype=torch.bfloat16):
    """a (M, K) @ b (K, N) -> (M, N), b row-major as stored (data gradient g @ W)."""
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, device=a.device, dtype=out_dtype)
    _call(a, b, c, None, M, N, K, a.stride(0), b.stride(0), N, 0, 1, 1)
    return c


def auto_splits(Kd, M, N, resident=512, cap=28):
    """Split-K factor of a weight gradient.  The (tile, K slice) workgroups should fill WHOLE rounds of the
    512 workgroups the chip holds at once (2 x 256 CUs): measured at FastVim-B, in_proj 4 slices (576 workgroups,
    1.1 rounds) 248 us vs 7 slices (1008, 2.0 rounds) 210 us; out_proj 8 slices 132 us vs 7 slices 95 us.  Among
    the factors that divide the K tiles, take the best round efficiency, then the fewest slices (every slice
    writes and re-reads an (M, N) fp32 partial); each slice is whole 64-deep K tiles."""
    if Kd % 64:
        return 1
    tiles = -(-M // 128) * -(-N // 128)
    kt = Kd // 64
    best, best_eff = 1, 0.0
    for s in range(1, min(cap, kt) + 1):
        if kt % s:
            continue
        w = tiles * s
        eff = w / (-(-w // resident) * resident)
        if eff > best_eff + 0.02:
            best, best_eff = s, eff
    return best


def gemm_tn(x, y, splits=1, out=None, accumulate=False, defer=True):
    """x (Kd, M)^T @ y (Kd, N) -> (M, N) fp32: weight gradient, reduction over the leading (token) dim
    cut into `splits` slices whose fp32 partials are summed in fixed order (deterministic split-K)."""
    Kd, M = x.shape
    N = y.shape[1]
    if splits is None:
        splits = auto_splits(Kd, M, N)
    while splits > 1 and (Kd % (splits * 64)):
        splits //= 2
    part = torch.empty(splits, M, N, device=x.device, dtype=torch.float32)
    _call(x, y, part, None, M, N, Kd, x.stride(0), y.stride(0), N, 1, 1, splits)
    if out is not None:
        # defer=False keeps the reduction on the stream the GEMM ran on (weight-gradient side stream)
        reduce_partials(part, splits, out=out, accumulate=accumulate, defer=defer)
        return None
    return part[0] if splits == 1 else reduce_partials(part, splits)


def grouped_splits(Kd, target=None, M=None, N=None):
    """Split-K factor inside a grouped launch: the queue of workgroups is long whatever the factor, so it trades the
    length of one workgroup's K loop and the rounds of workgroups the launch needs against the fp32 partial traffic
    (every slice writes and re-reads an (M, N) partial: at 7 slices 171 MB per FastVim-T step, 2.4 GB per FastVim-B
    step).  The largest divisor of the K tiles not above ``target``: 4 for outputs below 512 x 512 (FastVim-T: 5.90 ->
    5.87 ms per step against 7; 2 and 14 are slower, and so is a per-problem factor that evens out the workgroup
    counts) -- 6 uneven slices for the projection-sized ones among them since round 3 --, 2 below 1024 x 1024, 1 above
    (FastVim-B: 32.5 -> 32.1 ms)."""
    if target is None:
        if M is None or N is None or M * N < 512 * 512:
            target = 4
            # round 3: SIX slices for the FastVim-T projections (768 x 192, 192 x 384), the last one shorter when six does
            # not divide the K tiles (392 = 5 x 66 + 62) -- same box, three passes: 5.80 -> 5.72 ms per step (4 / 4: 5.80,
            # 6 / 4: 5.74, 6 / 6: 5.72, 7 / 7: 5.84, 8 / 8: 5.80, 14 / 7: 5.93; profiles/r03_sweep_wgrad_splits_t.log)
            if M is not None and N is not None and M * N >= 192 * 192 and Kd % 64 == 0 and Kd // 64 >= 6 * 14:
                per = -(-(Kd // 64) // 6)
                if -(-(Kd // 64) // per) == 6:
                    return 6
        else:
            target = 2 if M * N < 1024 * 1024 else 1
    if Kd % 64:
        return 1
    kt = Kd // 64
    best = 1
    for s_ in range(1, min(target, kt) + 1):
        if kt % s_ == 0:
            best = s_
    return best


def gemm_tn_grouped(jobs, reduce=True):
    """jobs: list of (x (Kd, M) bf16, y (Kd, N) bf16, out flat fp32 view of (M, N), splits).  ONE launch per 40
    problems computes every problem's split-K partials; the partials are then summed (deferred, fixed order) into
    ``out`` (accumulate).  ``reduce=False``: the sums are left to the caller -- returns [(partials, splits, out)] (a launch
    on a second stream must be joined before anything reads the partials)."""
    if not jobs:
        return []
    # one launch = one tile shape (the C side takes the shape that pads the launch's problems least: 128 x 192 for
    # outputs 192 wide, 192 x 128 for outputs 192 high, 256 x 256 when every output is a large multiple of it, else
    # 128 x 128): problems are grouped by the shape that suits them, and a problem that does not care (x_proj: 44 x 384)
    # joins the largest 128-row group it ties with
    classes = {}
    for j in jobs:
        classes.setdefault(_tile_class(j[0].shape[1], j[1].shape[1], j[0].shape[0]), []).append(j)
    if len(classes) > 1:
        todo = []
        loose = classes.pop(0, [])
        for j in loose:
            M, N = j[0].shape[1], j[1].shape[1]
            ties = [c for c in classes if c < 3 and _padded(M, N, c) == _padded(M, N, 0)]
            if ties:
                classes[max(ties, key=lambda c: len(classes[c]))].append(j)
            else:
                classes.setdefault(0, []).append(j)
        for c in sorted(classes):
            todo += _gemm_tn_grouped_one(classes[c], reduce, c)
        return todo
    return _gemm_tn_grouped_one(jobs, reduce, next(iter(classes)))


_TILES = ((128, 128), (128, 192), (192, 128), (256, 256), (256, 192), (192, 256))


def _padded(M, N, c):
    bm, bn = _TILES[c]
    return -(-M // bm) * bm * -(-N // bn) * bn


def _tile_class(M, N, Kd=None):
    """Index into _TILES of the tile shape that pads an (M, N) output least; ties go to 128 x 128.  ``Kd`` (tokens): the
    8-wave S-width tiles pay from long K loops on (FastChannelVim-S, 100 352 tokens: step 44.35 -> 43.51 ms; FastVim-S at
    224 px, 25 088 tokens: 12.55 -> 12.62, stays on 128 x 128; profiles/r05_ab_wgrad_s_width_tiles.log)."""
    if M % 256 == 0 and N % 256 == 0 and M * N >= 512 * 512:
        return 3          # large outputs (FastVim-B: 3072 x 768, 768 x 1536): 256 x 256 tiles on 8 waves halve the L2 traffic
    # S-width outputs (1536 x 384, 384 x 768): 256 x 192 / 192 x 256 tiles on 8 waves -- the C dispatcher's own rule
    # (fv_gemm_bf16_tn_grouped_wide8, K threshold included), so that the class the problems are grouped and split by is
    # the tile the launch runs on
    w8 = L.lib().fv_gemm_bf16_tn_grouped_wide8(L.i32(M), L.i32(N), L.i32(Kd if Kd is not None else 1 << 30))
    if w8:
        return int(w8)
    best = 0
    for c in (1, 2):
        if _padded(M, N, c) < _padded(M, N, best):
            best = c
    return best


DIRECT_ACC = True      # False: every grouped problem writes partials that a reduction launch adds to the gradient
_SLOTS = (512, 512, 512, 256, 256, 256)     # workgroups of the grouped kernel the chip holds at once, per tile class (two per CU;
                                  # one of the 8-wave 256 x 256 tiles)
FILL = True      # False: the factors of grouped_splits whatever the group (a segmented step then sums every weight
                 # gradient in the order of the single-graph step: tests/test_pipeline_gpu.py)


def fill_splits(jobs, c):
    """Split-K factors of one launch, raised where the launch would leave the chip part empty.  A group of all the
    blocks (the single-graph step) is 576-768 workgroups at the factors of ``grouped_splits`` and is left alone; a group
    of one backward segment of the data-parallel step (6 blocks: 144-192 workgroups, each walking a quarter of K) runs
    at half the rate per problem.  The work of the launch, in K tiles over all output tiles, spread over the 512 slots
    gives the K tiles one workgroup should walk; a problem is cut into the largest whole number of slices that does
    not go below that (and not below 14 tiles).  Measured, 6 blocks at FastVim-T: in_proj problems 156 -> 97 us,
    out_proj + x_proj 152 -> 102 us (tools/probe/wgrad_fill.py)."""
    bm, bn = _TILES[c]
    if c >= 4 and FILL:
        return _fill_splits_wide8(jobs, bm, bn)
    work = 0
    for x, y, out, sp in jobs:
        if x.shape[0] % 64 or not FILL:
            return [j[3] for j in jobs]
        work += -(-x.shape[1] // bm) * -(-y.shape[1] // bn) * (x.shape[0] // 64)
    per_slot = max(work / _SLOTS[c], 14.0)
    out_sp = []
    for x, y, out, sp in jobs:
        kt = x.shape[0] // 64
        best = sp
        for s_ in range(sp + 1, int(kt / per_slot) + 1):
            if kt % s_ == 0:
                best = s_
        out_sp.append(best)
    return out_sp


def _fill_splits_wide8(jobs, bm, bn):
    """One split-K factor for a launch of 8-wave tiles (one workgroup per CU, 256 slots): the launch runs in whole rounds
    of 256 workgroups, so the factor is the divisor of the K tiles that minimises  rounds x K tiles per workgroup  plus the
    partial traffic it adds (one K tile of a 256 x 192 tile ~ 1.95 us on a CU; an (M, N) fp32 partial written and re-read
    ~ M N 8 bytes at 4.5 TB/s).  FastChannelVim-S in_proj (24 problems x 12 tiles, 1568 K tiles): 2 slices = 2.25 rounds
    (3 paid), 8 slices = 9 rounds exactly."""
    kts = {x.shape[0] // 64 for x, _, _, _ in jobs}
    if len(kts) != 1 or any(x.shape[0] % 64 for x, _, _, _ in jobs):
        return [j[3] for j in jobs]
    kt = kts.pop()
    tiles = sum(-(-x.shape[1] // bm) * -(-y.shape[1] // bn) for x, y, _, _ in jobs)
    part_us = sum(x.shape[1] * y.shape[1] * 8 for x, y, _, _ in jobs) / 4.5e6
    base = min(j[3] for j in jobs)
    best, best_cost = base, None
    for s_ in range(1, kt // 14 + 1):
        if kt % s_:
            continue
        cost = -(-tiles * s_ // 256) * (kt // s_) * 1.95 + part_us * s_
        if best_cost is None or cost < best_cost:
            best, best_cost = s_, cost
    return [best for _ in jobs]


def _gemm_tn_grouped_one(jobs, reduce, c=None):
    k = len(jobs)
    if c is None:
        c = _tile_class(jobs[0][0].shape[1], jobs[0][1].shape[1], jobs[0][0].shape[0]) if k == 1 else 0
    jobs = [(x, y, out, s_) for (x, y, out, _), s_ in zip(jobs, fill_splits(jobs, c))]
    parts = []
    direct = []

    def _two_tiles(Kd, sp):      # the C side's rule for the phased 256 x 256 form: whole slices of at least two 64-deep K tiles
        kps = -(-(-(-Kd // sp)) // 64) * 64
        return kps >= 128 and Kd % kps == 0

    # in-place accumulation exists in the phased 256 x 256 kernel only, and the C side picks the kernel for the LAUNCH: one
    # problem whose slices are a single K tile sends the whole launch to the other kernel, so the flag is a launch property
    launch_phased = c == 3 and all(_two_tiles(x.shape[0], sp) for x, _, _, sp in jobs)
    for x, y, out, sp in jobs:
        Kd, M = x.shape
        N = y.shape[1]
        assert x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and x.stride(1) == 1 and y.stride(1) == 1
        assert out.numel() == M * N and out.dtype == torch.float32
        # one K slice of a large output (FastVim-B: 3072 x 768, 768 x 1536) is added to the gradient by the GEMM itself:
        # the (1, M, N) partial and the launch that summed it were 1 GB of traffic per step for nothing
        d = (DIRECT_ACC and launch_phased and sp == 1 and Kd % 64 == 0 and Kd >= 128 and out.is_contiguous()
             and out.data_ptr() % 16 == 0)
        direct.append(d)
        parts.append(out.view(1, M, N) if d else torch.empty(sp, M, N, device=x.device, dtype=torch.float32))
    P = ctypes.c_void_p
    xs = (P * k)(*[j[0].data_ptr() for j in jobs])
    ys = (P * k)(*[j[1].data_ptr() for j in jobs])
    ps = (P * k)(*[p_.data_ptr() for p_ in parts])
    I = ctypes.c_int
    Kds = (I * k)(*[j[0].shape[0] for j in jobs])
    Ms = (I * k)(*[j[0].shape[1] for j in jobs])
    Ns = (I * k)(*[j[1].shape[1] for j in jobs])
    sps = (I * k)(*[-1 if d else j[3] for j, d in zip(jobs, direct)])
    ldx = (I * k)(*[j[0].stride(0) for j in jobs])       # rows may be padded (a column slice of a wider buffer)
    ldy = (I * k)(*[j[1].stride(0) for j in jobs])
    rc = L.lib().fv_gemm_bf16_tn_grouped_ld(xs, ys, ps, Kds, Ms, Ns, ldx, ldy, sps, L.i32(k), L.stream_of(jobs[0][0]))
    L.check(rc, "gemm_bf16_tn_grouped")
    todo = [(part, sp, out) for (x, y, out, sp), part, d in zip(jobs, parts, direct) if not d]
    if not reduce:
        return todo
    for part, sp, out in todo:
        reduce_partials(part, sp, out=out, accumulate=True)
    return []
