"""GPU parity of the bf16 MFMA GEMM (fv_gemm_bf16) against fp64 matmul on the same bf16 inputs;
asymmetric integer-ish data would hide nothing: random data + exact-ish tolerance (fp32 accumulate)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b):
    return a.double().cpu() @ b.double().cpu()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 192, 192), (1000, 768, 192), (392, 384, 768),
                                   (25088, 192, 384), (130, 72, 40)])
def test_gemm_nt_nn(M, N, K):
    from fastvim_amd.gemm import gemm_nn, gemm_nt
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    ref = _ref(a, w.t())
    tol = 2.0 ** -7 * ref.abs().max().item()
    c = gemm_nt(a, w)
    assert (c.double().cpu() - ref).abs().max().item() <= tol
    c32 = gemm_nt(a, w, bias=bias, out_dtype=torch.float32)
    assert (c32.double().cpu() - (ref + bias.double().cpu())).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    b = w.t().contiguous()                     # (K, N) row-major
    c2 = gemm_nn(a, b, out_dtype=torch.float32)
    assert (c2.double().cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(25088, 768, 192),     # FastVim-T in_proj forward: 942 tiles of 160 x 128 (two rounds) instead of 1 176 of 128 x 128
                                   (25000, 768, 192),     # ragged last tile (40 of its 160 rows)
                                   (24999, 384, 128),     # ... and an odd row count
                                   (32768, 768, 192)])    # no round saved: stays on 128 x 128
def test_gemm_tall_tiles_that_save_a_round(M, N, K):
    """Forward GEMMs with a short K loop take 160 x 128 tiles where that saves a round of resident workgroups (the bf16
    epilogue's odd 16-row half slab): against fp64 at the bf16 rounding of the result, and row for row against the same
    product computed in two row blocks that do NOT take the tall tiles."""
    from fastvim_amd.gemm import gemm_nt
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    ref = _ref(a, w.t())
    c = gemm_nt(a, w)
    assert (c.double().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    half = 2048                                    # M < 4096: the 128 x 128 kernel
    for r0 in (0, M - half):
        assert torch.equal(c[r0:r0 + half], gemm_nt(a[r0:r0 + half], w))
    # the data-gradient form (B as stored, transposing reads) follows the same rule
    from fastvim_amd.gemm import gemm_nn
    b = w.t().contiguous()
    cn = gemm_nn(a, b)
    assert (cn.double().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    for r0 in (0, M - half):
        assert torch.equal(cn[r0:r0 + half], gemm_nn(a[r0:r0 + half], b))


@pytest.mark.parametrize("kind,M,N,K", [("nt", 100352, 1536, 384),    # FastChannelVim-S in_proj forward: 2352 tiles of 256 x 256
                                        ("nt", 25088, 3072, 768),     # FastVim-B in_proj forward
                                        ("nt", 65500, 1024, 384),     # ragged last row tile
                                        ("nt", 40000, 2048, 448),     # odd number of K tiles
                                        ("nn", 131072, 768, 3072),    # FastVim-B 2048 px in_proj data gradient (B as stored)
                                        ("nn", 70000, 1024, 512),     # ragged rows
                                        ("nn", 40000, 2048, 576)])    # odd number of K tiles
def test_gemm_phased_256_form(kind, M, N, K):
    """The phased 256 x 256 kernel (gemm_nt256pp_kernel: persistent workgroups, LDS-DMA in flight across barriers and across
    output tiles, counted waits) takes forward
    (K-contiguous weight) and data-gradient (weight as stored, transposing LDS reads) bf16 GEMMs of four rounds of tiles
    and more.  Against fp64 on the same inputs, against the fp32-output call of the same product (which runs the per-tile
    kernel; both accumulate in K order: the bf16 roundings must be equal), and bit-stable over repeated launches (a
    staging race shows as a tile that differs between launches)."""
    from fastvim_amd.gemm import gemm_nn, gemm_nt
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") if kind == "nt" else torch.randn(K, N, device="cuda")).bfloat16()
    fn = gemm_nt if kind == "nt" else gemm_nn
    c = fn(a, w)
    c32 = fn(a, w, out_dtype=torch.float32)
    assert torch.equal(c, c32.bfloat16())
    for _ in range(10):
        assert torch.equal(fn(a, w), c)
    rows = torch.randint(0, M, (512,), device="cuda")
    rows[:4] = torch.tensor([0, 255, M - 1, M - 2], device="cuda")
    wt = w.double().cpu().t() if kind == "nt" else w.double().cpu()
    ref = a[rows].double().cpu() @ wt
    assert (c[rows].double().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(25088, 768, 192), (25088, 192, 384), (25088, 192, 768), (25088, 384, 192),
                                   (8200, 192, 128), (12345, 384, 192)])
def test_gemm_streaming_form(M, N, K):
    """bf16-output forward (NT) and data-gradient (NN) GEMMs at the FastVim-T shapes go through the persistent
    streaming kernel (gemm_stream_kernel: M >= 8192, N % 192 == 0, N * K <= 192 * 768): trimmed tile heights, tiles
    ending mid-ring, ragged last tile (12345 rows), bias in the epilogue.  It accumulates in the same K order as
    the per-tile kernel, so the two must agree bit for bit (the fp32-output path always runs the per-tile kernel)."""
    from fastvim_amd.gemm import gemm_nn, gemm_nt
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda")
    ref = _ref(a, w.t())
    tol = 2.0 ** -7 * ref.abs().max().item()
    c = gemm_nt(a, w)
    assert (c.double().cpu() - ref).abs().max().item() <= tol
    cb = gemm_nt(a, w, bias=bias)
    assert (cb.double().cpu() - (ref + bias.double().cpu())).abs().max().item() <= tol
    b = w.t().contiguous()
    c2 = gemm_nn(a, b)
    assert (c2.double().cpu() - ref).abs().max().item() <= tol
    # the fp32-output path runs the per-tile kernel: rounding its result to bf16 must reproduce the streamed one
    c32 = gemm_nt(a, w, out_dtype=torch.float32)
    assert torch.equal(c32.bfloat16(), c)
    assert torch.equal(gemm_nn(a, b, out_dtype=torch.float32).bfloat16(), c2)
    # A as a column slice of a wider buffer (row stride > K), B as a row slice of a taller one
    wide = torch.randn(M, K + 64, device="cuda").bfloat16()
    wide[:, :K] = a
    tall = torch.randn(K + 16, N, device="cuda").bfloat16()
    tall[:K] = b
    assert torch.equal(gemm_nn(wide[:, :K], tall[:K]), c2)


@pytest.mark.parametrize("Kd,M,N,splits", [(64, 128, 128, 1), (1024, 192, 384, 4), (25088, 768, 192, 16),
                                           (25088, 192, 384, 8), (640, 72, 40, 2)])
def test_gemm_tn_splitk(Kd, M, N, splits):
    from fastvim_amd.gemm import gemm_tn
    torch.manual_seed(Kd + M)
    x = torch.randn(Kd, M, device="cuda").bfloat16()
    y = torch.randn(Kd, N, device="cuda").bfloat16()
    ref = _ref(x.t(), y)
    c = gemm_tn(x, y, splits=splits)
    assert (c.double().cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    acc = torch.ones(M, N, device="cuda")
    gemm_tn(x, y, splits=splits, out=acc, accumulate=True)
    assert (acc.double().cpu() - 1.0 - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    c2 = gemm_tn(x, y, splits=splits)
    assert torch.equal(c, c2)                  # deterministic


def test_gemm_strided_views():
    """operands that are column slices of wider buffers (xz halves, x_dbl parts)"""
    from fastvim_amd.gemm import gemm_nt, gemm_tn
    torch.manual_seed(0)
    big = torch.randn(512, 768, device="cuda").bfloat16()
    a = big[:, 384:]                            # (512, 384) with row stride 768
    w = torch.randn(192, 384, device="cuda").bfloat16()
    ref = _ref(a, w.t())
    assert (gemm_nt(a, w, out_dtype=torch.float32).double().cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    y = torch.randn(512, 64, device="cuda").bfloat16()
    ref2 = _ref(a.t(), y)
    assert (gemm_tn(a, y).double().cpu() - ref2).abs().max().item() <= 2e-4 * ref2.abs().max().item()


def test_grouped_weight_gradients_match_single_launches():
    """fv_gemm_bf16_tn_grouped: several x^T y problems of different shapes and split factors in one launch (more
    than 16 problems -> two launches) against fp64 and, for equal split factors, bit for bit against gemm_tn."""
    from fastvim_amd.gemm import gemm_tn, gemm_tn_grouped
    torch.manual_seed(0)
    shapes = [(1792, 768, 192, 7), (1792, 192, 384, 4), (896, 192, 768, 14), (640, 64, 40, 1), (1280, 136, 8, 5)] * 4
    jobs, refs, singles = [], [], []
    for Kd, M_, N_, sp in shapes:
        x = torch.randn(Kd, M_, device="cuda").bfloat16()
        y = torch.randn(Kd, N_, device="cuda").bfloat16()
        out = torch.zeros(M_ * N_, device="cuda")
        jobs.append((x, y, out, sp))
        refs.append(x.double().t() @ y.double())
        singles.append(gemm_tn(x, y, splits=sp))
    gemm_tn_grouped(jobs)
    from fastvim_amd.mixer_ops import flush_reductions
    flush_reductions()
    for (x, y, out, sp), ref, single in zip(jobs, refs, singles):
        got = out.view(ref.shape)
        assert (got.double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
        assert torch.equal(got, single)


def test_grouped_weight_gradients_tile_classes():
    """One grouped call mixing every tile-shape class (gemm.py groups them; the C side picks 128x192 / 192x128 /
    256x256 / 128x128 per launch) and padded x_proj-like operands: every problem against fp64 and, bit for bit, against
    the single-launch split-K GEMM (same K order whatever the tile)."""
    from fastvim_amd.gemm import _tile_class, gemm_tn, gemm_tn_grouped
    from fastvim_amd.mixer_ops import flush_reductions
    torch.manual_seed(1)
    shapes = [(1792, 768, 192, 7), (1792, 192, 384, 7), (1792, 3072, 768, 4), (1792, 768, 1536, 4), (896, 44, 384, 7),
              (896, 80, 1536, 7), (640, 128, 128, 5)]
    assert sorted({_tile_class(M_, N_) for _, M_, N_, _ in shapes}) == [0, 1, 2, 3]
    jobs, refs, singles = [], [], []
    for Kd, M_, N_, sp in shapes * 2:
        Mp = (M_ + 7) // 8 * 8
        xb = torch.randn(Kd, Mp, device="cuda").bfloat16()
        xb[:, M_:] = 0
        x = xb[:, :M_]                                   # a column slice of a padded buffer when M_ % 8 != 0
        y = torch.randn(Kd, N_, device="cuda").bfloat16()
        out = torch.zeros(M_ * N_, device="cuda")
        jobs.append((x, y, out, sp))
        refs.append(x.double().t() @ y.double())
        singles.append(gemm_tn(xb, y, splits=sp)[:M_] if Mp != M_ else gemm_tn(x, y, splits=sp))
    gemm_tn_grouped(jobs)
    flush_reductions()
    for (x, y, out, sp), ref, single in zip(jobs, refs, singles):
        got = out.view(ref.shape)
        assert (got.double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
        assert torch.equal(got, single)


def test_grouped_weight_gradients_s_width_tiles():
    """S-width outputs (in_proj 1536 x 384, out_proj 384 x 768: FastVim-S, FastChannelVim-S) run on 256 x 192 / 192 x 256
    tiles of 8 waves, each shape in a launch of its own with ONE split-K factor that fills whole rounds of workgroups
    (gemm.fill_splits): against fp64 and, bit for bit, against the single-launch split-K GEMM at that factor."""
    from fastvim_amd import gemm as G
    from fastvim_amd.mixer_ops import flush_reductions
    torch.manual_seed(2)
    # (the eight-wave tiles are taken from 50 000 tokens on -- the C dispatcher's rule, fv_gemm_bf16_tn_grouped_wide8, which
    #  the host's grouping asks too: one decider since round 6)
    shapes = [(50176, 1536, 384), (50176, 384, 768), (50176, 56, 768)]
    assert [G._tile_class(M_, N_) for _, M_, N_ in shapes] == [4, 5, 0]
    assert G._tile_class(1536, 384, 25088) == 0 and G._tile_class(1536, 384, 100352) == 4      # long K loops only
    jobs, refs = [], []
    for Kd, M_, N_ in shapes * 2:
        x = torch.randn(Kd, M_, device="cuda").bfloat16()
        y = torch.randn(Kd, N_, device="cuda").bfloat16()
        jobs.append((x, y, torch.zeros(M_ * N_, device="cuda"), G.grouped_splits(Kd, M=M_, N=N_)))
        refs.append(x.double().t() @ y.double())
    G.gemm_tn_grouped(jobs)
    flush_reductions()
    for c in (4, 5):
        mine = [j for j in jobs if G._tile_class(j[0].shape[1], j[1].shape[1], j[0].shape[0]) == c]
        sps = G.fill_splits(mine, c)
        assert len(set(sps)) == 1 and (50176 // 64) % sps[0] == 0
        for (x, y, out, _), sp in zip(mine, sps):
            assert torch.equal(out.view(x.shape[1], y.shape[1]), G.gemm_tn(x, y, splits=sp))
    for (x, y, out, _), ref in zip(jobs, refs):
        assert (out.view(ref.shape).double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_grouped_weight_gradients_accumulate_in_place():
    """Large outputs with ONE K slice (FastVim-B: 3072 x 768, 768 x 1536) are added to the gradient by the grouped GEMM
    itself (splits = -1 of fv_gemm_bf16_tn_grouped: no partial, no reduction launch): bit for bit what the partial +
    fv_reduce_partials route gives, on a gradient that already holds a first backward pass, and mixed in one call with a
    problem that keeps the partial route."""
    from fastvim_amd import gemm as G
    from fastvim_amd.mixer_ops import flush_reductions
    torch.manual_seed(2)
    shapes = [(1792, 3072, 768, 1), (1792, 768, 1536, 1), (1792, 768, 192, 7), (256, 512, 512, 1)]
    xs = [(torch.randn(Kd, M_, device="cuda").bfloat16(), torch.randn(Kd, N_, device="cuda").bfloat16()) for Kd, M_, N_, _ in shapes]
    first = [torch.randn(M_ * N_, device="cuda") for _, M_, N_, _ in shapes]
    outs = {}
    for mode in (True, False):
        G.DIRECT_ACC = mode
        try:
            out = [f.clone() for f in first]
            G.gemm_tn_grouped([(x, y, o, sp) for (x, y), o, (_, _, _, sp) in zip(xs, out, shapes)])
            flush_reductions()
            outs[mode] = out
        finally:
            G.DIRECT_ACC = True
    for a, b, (x, y), f in zip(outs[True], outs[False], xs, first):
        assert torch.equal(a, b)
        ref = f.double() + (x.double().t() @ y.double()).reshape(-1)
        assert (a.double() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    # the C entry refuses the in-place form where it is not built (an output that is not a multiple of 256 x 256)
    from fastvim_amd import _lib as L
    import ctypes
    x, y = torch.randn(256, 192, device="cuda").bfloat16(), torch.randn(256, 384, device="cuda").bfloat16()
    o = torch.zeros(192 * 384, device="cuda")
    P, I = ctypes.c_void_p, ctypes.c_int
    rc = L.lib().fv_gemm_bf16_tn_grouped_ld((P * 1)(x.data_ptr()), (P * 1)(y.data_ptr()), (P * 1)(o.data_ptr()), (I * 1)(256),
                                            (I * 1)(192), (I * 1)(384), (I * 1)(192), (I * 1)(384), (I * 1)(-1), L.i32(1),
                                            L.stream_of(x))
    assert rc != 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 768, 192), (392, 44, 384), (25088, 192, 384), (130, 72, 41),
                                   (7, 5, 3), (128, 1000, 192), (3584, 80, 1536), (8, 1000, 768)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_f32_mfma_all_layouts(M, N, K, dt):
    """fv_gemm_f32 (v_mfma_f32_16x16x4_f32; csrc/gemm_f32.hip) against fp64 matmul of the same operands: the three
    layouts the projections use (x W^T, g W, X^T Y) plus the fourth, fp32 and bf16 operands, bias epilogue, shapes with
    no alignment at all.  fp32 accumulate in K order: error <= K * 2^-23 of the sum of |products| (checked as 3e-6
    relative to the result scale -- a library GEMM is no tighter)."""
    from fastvim_amd.gemm import gemm_any_nn, gemm_any_nt, gemm_any_tn, gemm_f32
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, device="cuda").to(dt)
    w = torch.randn(N, K, device="cuda").to(dt)
    bias = torch.randn(N, device="cuda")
    ref = _ref(a, w.t())
    scale = max(1.0, ref.abs().max().item())
    tol = 3e-6 * scale * max(1.0, (K / 64) ** 0.5)
    c = gemm_any_nt(a, w, out_dtype=torch.float32)
    assert c.dtype == torch.float32 and (c.double().cpu() - ref).abs().max().item() <= tol
    cb = gemm_any_nt(a, w, bias=bias, out_dtype=torch.float32)
    assert (cb.double().cpu() - (ref + bias.double().cpu())).abs().max().item() <= tol
    c2 = gemm_any_nn(a, w.t().contiguous(), out_dtype=torch.float32)                     # B (K, N) as stored
    assert (c2.double().cpu() - ref).abs().max().item() <= tol
    if dt == torch.bfloat16:                                                               # bf16 C: one rounding of the fp32 result
        c16 = gemm_any_nt(a, w)
        assert c16.dtype == torch.bfloat16 and torch.equal(c16, c.bfloat16())
    # X^T Y: x (Kd, M), y (Kd, N)
    x, y = a.t().contiguous(), w.t().contiguous()
    for sp in (1, 2):
        part = gemm_any_tn(x, y, sp)
        assert (part.sum(0).double().cpu() - ref).abs().max().item() <= tol
    # the fourth layout (A K-slow, B K-contiguous), and row strides larger than the extents
    ap = torch.zeros(K, M + 5, device="cuda", dtype=dt)
    ap[:, :M] = x
    wp = torch.zeros(N, K + 3, device="cuda", dtype=dt)
    wp[:, :K] = w
    c4 = torch.empty(M, N, device="cuda")
    gemm_f32(ap, wp, c4, None, M, N, K, M + 5, K + 3, N, 1, 0)
    assert (c4.double().cpu() - ref).abs().max().item() <= tol
    # deterministic: bitwise the same on a second launch
    assert torch.equal(gemm_any_nt(a, w, out_dtype=torch.float32), c)


def test_gemm_f32_batched_forms():
    from fastvim_amd.gemm import gemm_any_bnn, gemm_any_bnt, gemm_any_btn
    torch.manual_seed(3)
    a = torch.randn(2, 1792, 384, device="cuda")
    w = torch.randn(2, 44, 384, device="cuda")
    ref = torch.bmm(a.double().cpu(), w.double().cpu().transpose(1, 2))
    got = gemm_any_bnt(a, w)
    assert (got.double().cpu() - ref).abs().max().item() <= 1e-4
    g = torch.randn(2, 1792, 44, device="cuda")
    ref2 = torch.bmm(g.double().cpu(), w.double().cpu())
    assert (gemm_any_bnn(g, w).double().cpu() - ref2).abs().max().item() <= 1e-4
    ref3 = torch.bmm(g.double().cpu().transpose(1, 2), a.double().cpu())
    assert (gemm_any_btn(g, a.bfloat16()).double().cpu()
            - torch.bmm(g.double().cpu().transpose(1, 2), a.bfloat16().double().cpu())).abs().max().item() <= 2e-3
    assert (gemm_any_btn(g, a).double().cpu() - ref3).abs().max().item() <= 2e-3
