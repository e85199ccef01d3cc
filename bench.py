#!/usr/bin/env python3
"""Headline benchmark: FastVim-T 224x224 bs=128/GPU bf16 training step (fwd + loss + bwd
[+ gradient all-reduce] + AdamW) on MI355X, images/sec whole-job.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline":     the dominant hand-written kernel, timed live with HIP events on the launch stream,
                  against its algorithmic HBM bytes (DESIGN.md section "Kernels and rooflines");
  "cpu_baseline": the CPU oracle (a port of the reference's pure-PyTorch path, selective_scan_ref
                  included) timed on this host on a bounded sample of the same workload.
Synthetic data: ImageNet-shaped N(0,1) pixels, Mixup-like soft targets; random-init weights.
"""
import argparse
import json
import os
import sys
import time

# ROCm 7.2's HIP-graph "packet capture" path (pre-recorded AQL packets) replays some captured training steps wrongly:
# FastVim-T at 512 px bs=32 and the MAE step at bs >= 64 turn non-finite after a few replays while the same steps are
# finite eagerly and with this switch off.  Off costs nothing measurable (8.74 vs 8.74 ms at
# FastVim-T 224 px) -- it must be set before the HIP runtime initialises.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_LIB"):      # A/B probes (tools/probe/*_variants.sh): a scratch build of the kernel library
    import fastvim_amd._lib as _probe_lib
    _probe_lib.LIB_PATH = os.environ["PROBE_LIB"]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0


def build_model(name, img_size, drop_path, channels=8):
    if name == "C":      # FastChannelVim-S/16 (BASELINE configs[4]); HCS off so the token count is the config's L = 196*channels
        from fastvim_amd.models_channel_mamba_faster import (
            channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2 as chan_s)
        return chan_s(img_size=img_size, channels=channels, hcs=False, drop_path_rate=drop_path)
    if name == "V":      # Vim-T baseline (models/vim.py): un-pooled scan, middle class token -- the paper's comparison point
        from fastvim_amd.vim import vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2 as vim_t
        return vim_t(img_size=img_size, drop_path_rate=drop_path)
    if name == "CV":     # ChannelVim-S/16, the un-pooled channel baseline with a middle class token (cell_imaging/config/ChannelVimS.yaml:24)
        from fastvim_amd.models_channel_mamba import (
            channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2 as cvim_s)
        return cvim_s(img_size=img_size, channels=channels, hcs=False, drop_path_rate=drop_path)
    if name == "MV":     # Vim-B masked autoencoder, the un-pooled MAE baseline (mae/config/pretrain_VimB.yaml:22)
        from fastvim_amd.fastvim_mae import mae_vim_base_dec512d2b
        return mae_vim_base_dec512d2b(img_size=img_size)
    if name == "M":      # FastVim-B masked autoencoder, the reference's MAE pre-training model (mae/config/pretrain_FastVimB.yaml:25)
        from fastvim_amd.models_mae import mae_FastVim_base_dec512d2b
        return mae_FastVim_base_dec512d2b(img_size=img_size)
    from fastvim_amd import fastvim as fv
    factory = {"T": fv.FastVimT, "S": fv.FastVimS, "B": fv.FastVimB}[name]
    return factory(img_size=img_size, drop_path_rate=drop_path)


def param_groups(model, weight_decay):
    """AdamW groups of the reference recipe (imagenet_classification/utils.py:52-69): no decay for
    1-D tensors, ``_no_weight_decay`` params and ``no_weight_decay()`` names."""
    skip = model.no_weight_decay()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or n.endswith(".bias") or n in skip or getattr(p, "_no_weight_decay", False):
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


def soft_targets(batch, num_classes, gen, device, smoothing=0.1):
    """Mixup/label-smoothing shaped targets (supervised_imagenet.py:69-83 uses timm Mixup)."""
    y1 = torch.randint(0, num_classes, (batch,), generator=gen)
    y2 = torch.randint(0, num_classes, (batch,), generator=gen)
    lam = 0.8
    off = smoothing / num_classes
    t = torch.full((batch, num_classes), off)
    t.scatter_add_(1, y1[:, None], torch.full((batch, 1), lam * (1 - smoothing)))
    t.scatter_add_(1, y2[:, None], torch.full((batch, 1), (1 - lam) * (1 - smoothing)))
    return t.to(device)


# --------------------------------------------------------------------------- kernel roofline
IC_BYTES = 256 << 20           # MI355X Infinity Cache (memory-side); plus 8 x 4 MB of L2


def time_kernel(fns, iters=20, warm=3):
    """Device time per call.  ``fns``: one callable, or a LIST of callables that launch the same kernel on different
    operand sets -- the launches then cycle through the sets, so that a set has been evicted from the Infinity Cache by
    the time it comes round again (HBM-cold timing, what the kernel sees inside the training step, where ~15 GB pass
    between two launches of the same kernel).  The calls are captured once into a HIP graph of back-to-back launches
    (host / ctypes overhead out of the picture), tensors they return are kept alive during capture so every launch
    also WRITES its own memory, and the replay is bracketed by HIP events recorded on the launch stream."""
    if callable(fns):
        fns = [fns]
    n = len(fns)
    iters = n * max(1, -(-iters // n))
    for i in range(max(warm, 1)):
        fns[i % n]()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    keep = []
    # with a process group alive its watchdog thread polls events; a "global" capture turns that into a capture error
    #  -- N > 1 runs time their kernel rows on rank 0 with the other ranks parked in a barrier
    mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode=mode):
            for i in range(iters):
                keep.append(fns[i % n]())
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    # median of three timed replays: one replay now and then catches a stall that is not the kernel's (a cfg-3 scan row
    # once read 458 us against 52-54 us in every other run and 46.5 us in the step trace)
    times = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e))
    del keep
    return sorted(times)[1] * 1e-3 / iters   # seconds per launch (incl. its own reduce_partials, if any)


def rotating(fn, base, names, nbytes, cap=48):
    """Closures of ``fn(operand set)`` over enough copies of the tensors ``names`` of ``base`` that the copies together
    exceed the Infinity Cache + L2s by a margin (at least 3 sets): cold operands for ``time_kernel``."""
    n = min(cap, max(3, -(-(IC_BYTES + (96 << 20)) // max(int(nbytes), 1)) + 1))
    sets = [base] + [dict(base, **{k: base[k].clone() for k in names}) for _ in range(n - 1)]
    return [lambda s_=s_: fn(s_) for s_ in sets]


def stream_floor(B, L, d_in, dtype):
    """What PLAIN elementwise kernels get on HBM-cold tensors of the step's full-length size U = (B, L, d_in): a copy (1 read,
    1 write) and an add (2 reads, 1 write), operand sets rotated exactly like the kernel rows.  The reference point for the
    short HBM-bound kernels: at U = 19.3 MB a launch is 10-15 us long and its ramp and drain keep even a copy at 3.0-4.5
    TB/s of the 8 TB/s peak (6.3 TB/s is what a long read-only stream reaches)."""
    n = B * L * d_in
    e = 2 if dtype == torch.bfloat16 else 4
    base = {k: torch.randn(n, device="cuda").to(dtype) for k in ("a", "b", "c")}
    out = {}
    for name, fn, names, streams in (("copy_1r1w", lambda s: s["b"].copy_(s["a"]), ("a", "b"), 2),
                                     ("add_2r1w", lambda s: torch.add(s["a"], s["b"], out=s["c"]), ("a", "b", "c"), 3)):
        fns = rotating(fn, base, names, n * e * len(names))
        t = time_kernel(fns)
        out[name] = {"us": round(t * 1e6, 2), "MB": round(streams * n * e / 1e6, 2), "GBps": round(streams * n * e / t / 1e9, 1)}
        del fns
    return out


def floor_curve(dtype, sizes_mb=(8, 16, 32, 64, 112, 160)):
    """[(traffic MB, us)] of a PLAIN copy kernel (1 read + 1 write) on HBM-cold operands, total traffic as given: the
    price of merely moving a kernel row's algorithmic bytes in one launch on this box, same timing method as the rows."""
    e = 2 if dtype == torch.bfloat16 else 4
    pts = []
    for mb in sizes_mb:
        n = int(mb * 1e6 / 2 / e)
        base = {k: torch.randn(n, device="cuda").to(dtype) for k in ("a", "b")}
        fns = rotating(lambda s: s["b"].copy_(s["a"]), base, ("a", "b"), 2 * n * e)
        pts.append((2 * n * e / 1e6, time_kernel(fns) * 1e6))
        del fns, base
    return pts


def floor_us(curve, mb):
    """Linear interpolation of ``floor_curve`` (extrapolated with the end slopes)."""
    if mb <= curve[0][0]:
        return curve[0][1] * max(mb / curve[0][0], 0.5)          # small launches: the fixed part dominates
    for (x0, y0), (x1, y1) in zip(curve[:-1], curve[1:]):
        if mb <= x1:
            return y0 + (y1 - y0) * (mb - x0) / (x1 - x0)
    (x0, y0), (x1, y1) = curve[-2], curve[-1]
    return y1 + (y1 - y0) * (mb - x1) / (x1 - x0)


def add_floors(kt, dtype):
    """Per kernel row: ``floor_us`` = a plain copy moving the row's own algorithmic bytes (HBM-cold, this run) and
    ``floor_ratio`` = us / floor_us (1.0 = the row costs what moving its bytes costs; the verdict's bar is 1.25)."""
    curve = floor_curve(dtype)
    for v in kt.values():
        if "algorithmic_MB" in v and v.get("launches_per_step", 1) >= 1 and "us" in v:
            f = floor_us(curve, v["algorithmic_MB"])
            v["floor_us"] = round(f, 2)
            v["floor_ratio"] = round(v["us"] / f, 2)
    return [(round(a, 1), round(b, 2)) for a, b in curve]


def kernel_table(B, rows, cols, d, depth, dtype, tpp=1):
    """Time every hand-written full-length kernel of one mixer block at the benchmark shape and price it against its
    ALGORITHMIC bytes (formulas in DESIGN.md).  ``us`` is HBM-COLD (operand sets rotated past the Infinity Cache, as
    inside the step); ``us_warm`` is the same launch repeated on one operand set (cache-resident: an upper bound on
    what the kernel can do, not what the step sees)."""
    from fastvim_amd import mixer_ops as M
    from fastvim_amd.layernorm import layer_norm_fn
    dev = "cuda"
    d_in, L, R, N = 2 * d, rows * cols * tpp, -(-d // 16), 16
    prow = rows * tpp                        # pooled rows per image (channel-wise tokenization: one per channel token slot)
    e = 2 if dtype == torch.bfloat16 else 4
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, dt=dtype: torch.randn(*s, device=dev, generator=g).to(dt)
    T = {"xz": rn(B, L, 2 * d_in)}
    cw, cwb = rn(d_in, 4, dt=torch.float32) * 0.5, rn(d_in, 4, dt=torch.float32) * 0.5
    cb, cbb = rn(d_in, dt=torch.float32) * 0.1, rn(d_in, dt=torch.float32) * 0.1
    D, Db = torch.ones(d_in, device=dev), torch.ones(d_in, device=dev)
    lnw, lnb = torch.ones(d_in, device=dev), torch.zeros(d_in, device=dev)
    Wdt = rn(d_in, R, dt=torch.float32) * R ** -0.5
    bdt = torch.full((d_in,), -4.0, device=dev)
    A_log = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1).contiguous()
    T["xc"], T["skip"] = M.conv_pool_fwd(T["xz"], cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, tpp, D=D, D_b=Db)
    T["x_dbl"] = rn(2, B * prow, R + 2 * N)
    T["yc"] = M.scan_fwd(T["xc"], T["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log)
    _, T["mean"], T["rstd"] = M.combine_fwd(T["xz"], T["skip"], T["yc"], lnw, lnb, 1e-5, rows, cols, False, tpp=tpp)
    T["dg"] = rn(B, L, d_in)
    T["dxz"] = torch.empty_like(T["xz"])
    T["d_o"], T["dyc"], _ = M.combine_bwd(T["dg"], T["xz"], T["skip"], T["yc"], lnw, lnb, T["mean"], T["rstd"], T["dxz"],
                                          rows, cols, False, tpp=tpp)
    T["dxc"] = torch.randn(2, B, prow, d_in, device=dev, generator=g)
    T["hid"], T["res"] = rn(B, L, d), torch.randn(B, L, d, device=dev, generator=g)
    nw = torch.ones(d, device=dev)
    U = B * L * d_in * e                      # one full-length (B, L, d_in) tensor
    small = B * prow * d_in
    table = {      # name: (launch on an operand set, the set's tensors that rotate, algorithmic bytes, launches per block)
        "conv_pool_fwd": (lambda s: M.conv_pool_fwd(s["xz"], cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, tpp, D=D, D_b=Db),
                          ("xz",), 2 * U + 2 * small * e, 1),            # x read, skip written, xc written
        "scan_fwd": (lambda s: M.scan_fwd(s["xc"], s["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log),
                     ("xc", "x_dbl"), 2 * (small * e + B * prow * (R + 2 * N) * e + small * 4), 1),
        "combine_fwd": (lambda s: M.combine_fwd(s["xz"], s["skip"], s["yc"], lnw, lnb, 1e-5, rows, cols, False, tpp=tpp),
                        ("xz", "skip", "yc"), 3 * U + 2 * small * 4 + 2 * B * L * 4, 1),    # skip, z read; g written
        "combine_bwd": (lambda s: M.combine_bwd(s["dg"], s["xz"], s["skip"], s["yc"], lnw, lnb, s["mean"], s["rstd"],
                                                s["dxz"], rows, cols, False, tpp=tpp),
                        ("dg", "xz", "skip", "yc", "mean", "rstd", "dxz"),
                        5 * U + 3 * small * 4 + 2 * B * L * 4, 1),   # dg, z, skip read; dz, do written
        "scan_bwd": (lambda s: M.scan_bwd(s["xc"], s["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, s["dyc"]),
                     ("xc", "x_dbl", "dyc"),
                     2 * (small * e + B * prow * (R + 2 * N) * (e + 4) + small * 4) + small * 4, 1),
        "conv_pool_bwd": (lambda s: M.conv_pool_bwd(s["xz"], s["d_o"], s["dxc"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols,
                                                    False, 0, 1.0, tpp=tpp),
                          ("xz", "d_o", "dxc", "dxz"), 3 * U + 2 * small * 4, 1),
        "add_rmsnorm_fwd": (lambda s: layer_norm_fn(s["hid"], nw, None, residual=s["res"], eps=1e-5, prenorm=True,
                                                    residual_in_fp32=True, is_rms_norm=True),
                            ("hid", "res"), B * L * d * (2 * e + 8), 1),
    }
    # short pooled lengths in bf16 (configs 2 / 3 up to d_inner 768): the step runs x_proj + dt_proj + scan as ONE launch
    # (fv_mixer_xproj_scan_fwd) -- time that launch under the scan_fwd name, as the step trace does
    if dtype == torch.bfloat16:
        Wx2c = (rn(2, R + 2 * N, d_in, dt=torch.float32) * d_in ** -0.5).to(dtype)
        if M.xproj_scan_fwd(T["xc"], Wx2c, Wdt, bdt, A_log, Wdt, bdt, A_log) is not None:
            table["scan_fwd"] = (lambda s: M.xproj_scan_fwd(s["xc"], Wx2c, Wdt, bdt, A_log, Wdt, bdt, A_log), ("xc",),
                                 2 * (small * e + B * prow * (R + 2 * N) * e + small * 4) + Wx2c.numel() * e, 1)
    # the x_proj adjoint as the step issues it (short pooled lengths: chunk sum + dxc += dx_dbl @ Wx + the bf16 dx_dbl rows
    # the grouped weight-gradient launch reads; the weight gradient itself is in gemm_wgrad_grouped).  Algorithmic bytes:
    # the scan backward's chunk partials read once, dxc read and written, the fp32 weight, the bf16 rows written
    W_ = R + 2 * N
    folded = (dtype == torch.bfloat16 and tpp == 1 and M.scan_bwd_xproj_ok(T["xc"], Wdt, False, rows, cols, 1))
    if folded:
        # FastVim-T: the x_proj adjoint's data half runs INSIDE the scan backward (round 5) and the conv + pool adjoint
        # takes the pooled gradient as two addends -- timed as the step issues them.  Bytes: the scan row's + the second
        # addend (storage dtype) written once and read once + the fp32 x_proj weight
        Wx32 = rn(2, W_, d_in, dt=torch.float32) * d_in ** -0.5
        _, T["dxc2"], _, _ = M.scan_bwd_xproj(T["xc"], T["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, T["dyc"], Wx32[0], Wx32[1])
        sb = table["scan_bwd"]
        table["scan_bwd"] = (lambda s: M.scan_bwd_xproj(s["xc"], s["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, s["dyc"], Wx32[0], Wx32[1]),
                             sb[1], sb[2] + 2 * small * e + 2 * W_ * d_in * 4, 1)
        cb_ = table["conv_pool_bwd"]
        table["conv_pool_bwd"] = (lambda s: M.conv_pool_bwd(s["xz"], s["d_o"], s["dxc"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols,
                                                            False, 0, 1.0, dxc2=s["dxc2"]),
                                  cb_[1] + ("dxc2",), cb_[2] + 2 * small * e, 1)
    elif dtype == torch.bfloat16 and W_ in M.XPROJ_WIDTHS:
        Wx32 = rn(2, W_, d_in, dt=torch.float32) * d_in ** -0.5
        _, T["dxdbl_chunks"], _ = M.scan_bwd(T["xc"], T["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, T["dyc"], keep_chunks=True)
        nch = T["dxdbl_chunks"].shape[0]
        if nch < M._XPROJ_PRESUM:
            # as the flat training step issues it: where the bf16 matrix-core form is built (round 6, wide models) the
            # product is taken from the transposed bf16 shadow weight
            WxT = (Wx32.to(torch.bfloat16).transpose(1, 2).contiguous()
                   if M.xproj_bwd3_ok(B * prow, d_in, W_, dtype) else None)
            table["xproj_bwd"] = (lambda s: M.xproj_bwd(s["dxdbl_chunks"], s["xc"], Wx32[0], Wx32[1], s["dxc"], dw=False, Wx2_t=WxT),
                                  ("dxdbl_chunks", "dxc"),
                                  nch * 2 * B * prow * W_ * 4 + 2 * (2 * small * 4) + 2 * W_ * d_in * (2 if WxT is not None else 4)
                                  + 2 * B * prow * ((W_ + 7) // 8 * 8) * 2, 1)
    out = {}
    # the backward wrappers sum their per-block gradient partials right away when no flat gradient is attached; in
    # the training step those sums are deferred into reduce_partials_multi launches (timed by the step, not here), so
    # they are switched off while a row's own kernel is timed
    real_reduce = M.reduce_partials
    M.reduce_partials = lambda part, n, out=None, **kw: out if out is not None else part[0]
    try:
        timed = {}
        for name, (fn, names, nbytes, _) in table.items():
            fns = rotating(fn, T, names, nbytes)
            timed[name] = (time_kernel(fns), time_kernel(fns[0]), len(fns))
            del fns
    finally:
        M.reduce_partials = real_reduce
    for name, (fn, names, nbytes, per_block) in table.items():
        t, tw, nsets = timed[name]
        out[name] = {"us": round(t * 1e6, 2), "us_warm": round(tw * 1e6, 2), "operand_sets": nsets,
                     "algorithmic_MB": round(nbytes / 1e6, 3),
                     "GBps": round(nbytes / t / 1e9, 1), "launches_per_step": per_block * depth,
                     "us_per_step": round(t * 1e6 * per_block * depth, 1)}
    if dtype == torch.bfloat16:
        # the six projection GEMMs of a block (hand-written MFMA kernel, csrc/gemm_mfma.hip): HBM bytes AND
        # MFMA flops -- at FastVim-T widths (K or N = 192) they sit below the ridge, i.e. are HBM-bound
        from fastvim_amd.gemm import gemm_nn, gemm_nt
        Mt = B * L
        Gs = {"h2": rn(Mt, d), "g2": rn(Mt, d_in), "xz2": rn(Mt, 2 * d_in), "do2": rn(Mt, d)}
        h2, g2, xz2, do2 = Gs["h2"], Gs["g2"], Gs["xz2"], Gs["do2"]
        W_in, W_out = rn(2 * d_in, d), rn(d, d_in)
        gemms = {
            "gemm_in_proj_fwd": (lambda s: gemm_nt(s["h2"], W_in), ("h2",), Mt, 2 * d_in, d, 0),
            "gemm_out_proj_fwd": (lambda s: gemm_nt(s["g2"], W_out), ("g2",), Mt, d, d_in, 0),
            "gemm_out_proj_dgrad": (lambda s: gemm_nn(s["do2"], W_out), ("do2",), Mt, d_in, d, 0),
            "gemm_in_proj_dgrad": (lambda s: gemm_nn(s["xz2"], W_in), ("xz2",), Mt, d, 2 * d_in, 0),
        }
        # weight gradients run as grouped launches at the end of backward: the in_proj / out_proj
        # problems of ALL blocks are timed here exactly as the step issues them (gemm_tn_grouped: one tile-shape class
        # per launch, <= 40 problems each; reductions of the fp32 partials included).  Every problem has its OWN
        # operands -- the blocks' activations and gradients, 1.9 GB at FastVim-T -- when they fit 6 GiB: re-using one set
        # would keep it in the 256 MB Infinity Cache and time a different kernel
        from fastvim_amd.gemm import gemm_tn_grouped, grouped_splits
        from fastvim_amd.mixer_ops import flush_reductions
        sp_i, sp_o = grouped_splits(Mt, M=2 * d_in, N=d), grouped_splits(Mt, M=d, N=d_in)
        sp = sp_i
        per_block = Mt * (3 * d_in + 2 * d) * 2
        nset = depth if per_block * depth <= (6 << 30) else 1
        sets = [(xz2, h2, do2, g2)] + [(rn(Mt, 2 * d_in), rn(Mt, d), rn(Mt, d), rn(Mt, d_in)) for _ in range(nset - 1)]
        gis = [torch.zeros(2 * d_in * d, device=dev) for _ in range(depth)]
        gos = [torch.zeros(d * d_in, device=dev) for _ in range(depth)]
        group = []
        for i in range(depth):
            a, b_, c_, e_ = sets[i % nset]
            group += [(a, b_, gis[i], sp_i), (c_, e_, gos[i], sp_o)]

        def wgrad_group():
            gemm_tn_grouped(group)
            flush_reductions()
        from fastvim_amd.mixer_ops import _Deferred, defer_reductions
        was = _Deferred.enabled
        defer_reductions(True)          # as in the step: the partial sums of a group go out as ONE multi-reduction launch
        try:
            t = time_kernel(wgrad_group, iters=5)
        finally:
            defer_reductions(was)
        fl = depth * 2.0 * Mt * (2 * d_in * d + d * d_in)
        nbytes = depth * (2 * Mt * (2 * d_in + d + d + d_in) + 4 * ((2 * sp_i + 1) * 2 * d_in * d + (2 * sp_o + 1) * d * d_in))
        out["gemm_wgrad_grouped"] = {
            "us": round(t * 1e6, 2), "algorithmic_MB": round(nbytes / 1e6, 3), "GBps": round(nbytes / t / 1e9, 1),
            "TFLOPs": round(fl / t / 1e12, 1), "mfma_frac": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "launches_per_step": 1, "us_per_step": round(t * 1e6, 1), "split_k": sp, "problems": 2 * depth,
            "own_operands_per_problem": nset == depth}
        del sets, group

        def gemm_row(name, fn, names, base, nbytes, fl, launches):
            fns = rotating(fn, base, names, nbytes)
            t, tw = time_kernel(fns), time_kernel(fns[0])
            out[name] = {"us": round(t * 1e6, 2), "us_warm": round(tw * 1e6, 2), "operand_sets": len(fns),
                         "algorithmic_MB": round(nbytes / 1e6, 3), "GBps": round(nbytes / t / 1e9, 1),
                         "TFLOPs": round(fl / t / 1e12, 1), "mfma_frac": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                         "launches_per_step": launches, "us_per_step": round(t * 1e6 * launches, 1)}

        if d == 192:
            # inside a run of blocks (fastvim._run_layers_chained) out_proj runs fused with the next block's add + RMSNorm,
            # and the in_proj data gradient fused with the block's norm adjoint and the previous block's out_proj data
            # gradient: the rows for those GEMMs and for add_rmsnorm_fwd are then launched once per step (first block /
            # closing out_proj), these depth - 1 times
            import ctypes
            from fastvim_amd import _lib as L_
            lib = L_.lib()
            nw_, sc_ = torch.ones(d, device=dev), torch.ones(B, device=dev)
            F_ = dict(Gs, resid=torch.randn(Mt, d, device=dev, generator=g), rstd=torch.rand(Mt, device=dev, generator=g) + 0.5,
                      y=torch.empty(Mt, d, device=dev, dtype=dtype), ro=torch.empty(Mt, d, device=dev),
                      rs=torch.empty(Mt, device=dev), gg=torch.randn(Mt, d, device=dev, generator=g),
                      pw=torch.empty(lib.fv_gemm_bf16_dgrad_addnorm_blocks(L_.i32(Mt)), d, device=dev),
                      dg=torch.empty(Mt, d_in, device=dev, dtype=dtype))
            W_in_t = W_in.contiguous()                  # (2 d_in, d): K x N as stored

            def fused_fwd(s):
                L_.check(lib.fv_gemm_bf16_addnorm(L_.ptr(s["g2"]), L_.ptr(W_out), L_.ptr(s["resid"]), L_.ptr(nw_), L_.ptr(sc_),
                                                  L_.i32(L), L_.ptr(s["y"]), L_.ptr(s["ro"]), L_.ptr(s["rs"]), L_.i32(Mt),
                                                  L_.i32(d), L_.i32(d_in), ctypes.c_long(d_in), ctypes.c_long(d_in),
                                                  ctypes.c_float(1e-5), L_.stream_of(s["g2"])), "addnorm")

            def fused_bwd(s):       # with its second phase: the previous block's out_proj data gradient from the d x tile
                L_.check(lib.fv_gemm_bf16_dgrad_addnorm_bwd2(L_.ptr(s["xz2"]), L_.ptr(W_in_t), L_.ptr(s["gg"]), L_.ptr(s["resid"]),
                                                             L_.ptr(s["rstd"]), L_.ptr(nw_), L_.ptr(sc_), L_.i32(L), L_.ptr(s["y"]),
                                                             L_.ptr(s["ro"]), L_.ptr(s["pw"]), L_.i32(Mt), L_.i32(d),
                                                             L_.i32(2 * d_in), ctypes.c_long(2 * d_in), ctypes.c_long(d),
                                                             L_.ptr(W_out), L_.ptr(s["dg"]), L_.i32(d_in), ctypes.c_long(d_in),
                                                             L_.stream_of(s["xz2"])), "dgrad_addnorm_bwd")
            fuse_c = tpp == 1 and M.combine_out_proj_addnorm_ok(T["xz"], rows, cols, 1, d)
            if fuse_c:
                # round 5: combine is the A-tile producer of this launch from the third block on (fastvim._run_layers_chained)
                C_ = dict(T, resid=F_["resid"])
                C_["gbuf"], C_["mbuf"], C_["rbuf"] = M.combine_buffers(T["xz"], lnw)
                sc_b = torch.ones(B, device=dev)

                def fused_cfwd(s):
                    return M.combine_out_proj_addnorm(s["xz"], s["skip"], s["yc"], lnw, lnb, 1e-5, rows, cols, False,
                                                      (s["gbuf"], s["mbuf"], s["rbuf"]), W_out, s["resid"], nw_, sc_b, L, 1e-5)
                gemm_row("combine_out_proj_addnorm_fwd", fused_cfwd, ("xz", "skip", "yc", "resid", "gbuf"), C_,
                         3 * U + 2 * small * 4 + 2 * B * L * 4 + Mt * (d * (4 + 4 + e) + 4), 2.0 * Mt * d * d_in, depth - 2)
                out["combine_fwd"]["launches_per_step"] = 2
                out["combine_fwd"]["us_per_step"] = round(out["combine_fwd"]["us"] * 2, 1)
                del C_
            gemm_row("gemm_out_proj_addnorm_fwd", fused_fwd, ("g2", "resid", "y", "ro", "rs"), F_,
                     Mt * (d_in * e + d * (4 + 4 + e) + 4), 2.0 * Mt * d * d_in, 1 if fuse_c else depth - 1)
            fuse_cd = tpp == 1 and M.conv_pool_bwd_dgrad_ok(T["xz"], rows, cols, 1, d, False)
            gemm_row("gemm_in_proj_dgrad_addnorm_bwd", fused_bwd, ("xz2", "gg", "resid", "rstd", "y", "ro", "pw", "dg"), F_,
                     Mt * (2 * d_in * e + d * (4 + 4 + 4 + e) + 4 + d_in * e), 2.0 * Mt * d * 3 * d_in, 0 if fuse_cd else depth - 1)
            if fuse_cd:
                # round 6: the conv + pool adjoint is the A-tile producer of that launch in every chained block
                # (fv_mixer_conv_pool_bwd_dgrad); the two rows it replaces stay in the table with the launches they keep (the
                # first block's conv adjoint; no stand-alone data gradient + norm adjoint).  Algorithmic bytes: x, d_o read
                # and the x half of dxz written (3U) + pooled gradients (+ the second addend) + the z half of dxz read (U)
                # + the norm adjoint's rows (residual gradient in / out, saved input, d hidden) + the previous block's d g
                # written (U); the x half of dxz is NOT read back
                CD_ = dict(T, gg=F_["gg"], resid=F_["resid"], rstd=F_["rstd"])
                if "dxc2" not in CD_:
                    CD_["dxc2"] = torch.randn(2, B, prow, d_in, device=dev, generator=g).to(dtype)
                W_in_T = W_in.t().contiguous()

                def fused_cd(s):
                    return M.conv_pool_bwd_dgrad(s["xz"], s["d_o"], s["dxc"], s["dxc2"], cw, cb, cwb, cbb, D, Db, s["dxz"], rows, cols,
                                                 False, 1.0, W_in_T, s["gg"], s["resid"], s["rstd"], nw_, sc_, L, W2=W_out)
                gemm_row("conv_pool_bwd_dgrad_addnorm_bwd", fused_cd, ("xz", "d_o", "dxc", "dxc2", "dxz", "gg", "resid", "rstd"), CD_,
                         5 * U + 2 * small * 4 + 2 * small * e + Mt * (d * (4 + 4 + 4 + e) + 4), 2.0 * Mt * d * 3 * d_in, depth - 1)
                out["conv_pool_bwd"]["launches_per_step"] = 1
                out["conv_pool_bwd"]["us_per_step"] = out["conv_pool_bwd"]["us"]
                del CD_
            del F_
            for k_ in ("add_rmsnorm_fwd",):
                out[k_]["launches_per_step"], out[k_]["us_per_step"] = 2, round(out[k_]["us"] * 2, 1)
        for name, (fn, names, m_, n_, k_, splits) in gemms.items():
            nbytes = 2 * (m_ * k_ + n_ * k_) + (2 * m_ * n_ if not splits else 4 * m_ * n_ * (2 * splits + 1))
            gemm_row(name, fn, names, Gs, nbytes, 2.0 * m_ * n_ * k_, depth)
        if d == 192:
            for k_ in ("gemm_out_proj_fwd", "gemm_in_proj_dgrad", "gemm_out_proj_dgrad"):
                out[k_]["launches_per_step"], out[k_]["us_per_step"] = 1, out[k_]["us"]
    return out


# kernel-table row -> name prefix of the kernel in a rocprofv3 kernel trace of the graph-replayed step
TRACE_NAMES = {
    "conv_pool_fwd": "conv_pool_fwd_row_kernel", "scan_fwd": "xproj_scan_fwd_short_kernel", "combine_fwd": "combine_fwd_wave_kernel",
    "combine_bwd": "combine_bwd_wave_kernel", "scan_bwd": "scan_cl_bwd_short_kernel", "conv_pool_bwd": "conv_pool_bwd_row_kernel",
    "add_rmsnorm_fwd": "add_norm_fwd3_kernel", "gemm_out_proj_addnorm_fwd": "gemm_addnorm_kernel",
    "gemm_in_proj_dgrad_addnorm_bwd": "gemm_dgrad_addnorm_bwd_kernel", "gemm_in_proj_fwd": "gemm_bf16_kernel<0, 0, 2, 2, true, 4, 5>",
    "xproj_bwd": "xproj_bwd_mmb_kernel", "combine_out_proj_addnorm_fwd": "combine_out_proj_addnorm_kernel",
    "conv_pool_bwd_dgrad_addnorm_bwd": "conv_pool_bwd_dgrad_kernel",
}
PMC_TRAFFIC_JSON = "r06_v4_pmc_traffic.json"               # same script: three --pmc passes folded by tools/pmc_summary.py
STEP_TRACE_CSV = "r06_v4_graph_step_kernel_stats.csv"      # committed: bash tools/profile_step.sh r05_v4 (profiles/README.md)


def in_step_trace_us():
    """Average duration of each table row's kernel INSIDE the replayed FastVim-T step, from the committed rocprofv3
    kernel trace (profiles/): what the HBM-cold `us` of the table should agree with.  {} when the file is absent."""
    import csv
    path = os.path.join(ROOT, "profiles", STEP_TRACE_CSV)
    if not os.path.exists(path):
        return {}
    rows = list(csv.DictReader(open(path)))
    out = {}
    for key, prefix in TRACE_NAMES.items():
        for r in rows:
            if r["Name"].startswith(prefix):
                out[key] = round(float(r["AverageNs"]) / 1e3, 2)
                break
    return out


def step_trace_breakdown(top=40):
    """Every kernel of the replayed FastVim-T step from the committed rocprofv3 trace, per step: the rows of the kernel table
    AND what the table does not time by itself (partial-sum reductions, the bf16 d x_dbl rows, patch unfold, optimizer, the
    library nodes of the captured step) -- so that the rows sum to the step.  Steps in the trace = launches of the optimizer
    kernel.  None when the file is absent."""
    import csv
    path = os.path.join(ROOT, "profiles", STEP_TRACE_CSV)
    if not os.path.exists(path):
        return None
    rows = list(csv.DictReader(open(path)))
    steps = next((int(r["Calls"]) for r in rows if r["Name"].startswith("adamw_flat_kernel")), 0)
    if steps <= 0:
        return None
    table_prefixes = tuple(TRACE_NAMES.values()) + ("gemm_bf16_grouped_kernel", "gemm_stream_kernel", "gemm_bf16_kernel",
                                                    "add_norm_bwd3_kernel")
    out_rows, total, tail = [], 0.0, 0.0
    for r in rows:
        name = r["Name"]
        if name.startswith("__amd_rocclr_copyBuffer"):       # set-up copies (flattening the parameters): none inside a replayed step
            continue
        per_step = float(r["TotalDurationNs"]) / steps / 1e3
        total += per_step
        in_table = name.startswith(table_prefixes)
        if not in_table:
            tail += per_step
        out_rows.append({"kernel": name[:96], "launches_per_step": round(int(r["Calls"]) / steps, 2),
                         "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "us_per_step": round(per_step, 1), "in_kernel_table": in_table})
    out_rows.sort(key=lambda r: -r["us_per_step"])
    return {"file": STEP_TRACE_CSV, "steps_in_trace": steps, "kernel_us_per_step": round(total, 1),
            "tail_us_per_step_outside_kernel_table": round(tail, 1), "rows": out_rows[:top]}


def step_roofline(kt, ms_per_step):
    """Whole-step roofline from the kernel table: the ALGORITHMIC bytes of every row x its launches per step against the
    step time (HBM fraction), and the flops of the GEMM rows against the dense bf16 MFMA peak.  Rows the table does not
    carry (patch embed, head, loss, optimizer, reductions: ~3 % of the step) add bytes, so the fractions are slightly low."""
    nbytes = sum(v["algorithmic_MB"] * 1e6 * v["launches_per_step"] for v in kt.values())
    flops = sum(v["TFLOPs"] * 1e12 * v["us"] * 1e-6 * v["launches_per_step"] for v in kt.values() if "TFLOPs" in v)
    t = ms_per_step * 1e-3
    return {"algorithmic_GB": round(nbytes / 1e9, 3), "hbm_GBps": round(nbytes / t / 1e9, 1),
            "hbm_frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "gemm_TFLOP": round(flops / 1e12, 4),
            "gemm_TFLOPs": round(flops / t / 1e12, 1), "mfma_frac": round(flops / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
            "kernel_rows_us": round(sum(v["us_per_step"] for v in kt.values()), 1), "ms_per_step": round(ms_per_step, 3)}


def profile_stamp():
    """Where the committed-profile numbers of the line (roofline.traffic, us_in_step_trace) come from: file tags and the
    commit that last touched them -- they are NOT measured in this run and go stale when a kernel changes."""
    import subprocess
    st = {"pmc_traffic": PMC_TRAFFIC_JSON, "step_trace": STEP_TRACE_CSV, "measured_in_this_run": False}
    try:      # the commit the profiled tree was at, recorded in the file itself (the GPU box's copy of the repo has no .git)
        st["taken_at_commit"] = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_JSON))).get("taken_at_commit")
    except (OSError, ValueError):
        st["taken_at_commit"] = None
    try:
        r = subprocess.run(["git", "log", "-1", "--format=%h %cI", "--", os.path.join("profiles", PMC_TRAFFIC_JSON)],
                           cwd=ROOT, capture_output=True, text=True, timeout=10)
        h = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10)
        st["profile_commit"] = r.stdout.strip() or None
        st["head"] = h.stdout.strip() or None
    except Exception:
        st["profile_commit"] = st["head"] = None        # the GPU box's copy has no .git: the tags above still identify the files
    return st


def library_gemm_ceiling(M_, N_, K_, trans_b=True):
    """The vendor library (hipBLASLt through torch) on the same operands: a same-box sanity ceiling for the build's GEMMs,
    timed HERE only -- the product path never calls a library GEMM (tests/test_model_gpu.py::test_training_step_calls_no_library_gemm)."""
    a = torch.randn(M_, K_, device="cuda").bfloat16()
    sets = [{"a": a, "w": (torch.randn(N_, K_, device="cuda") if trans_b else torch.randn(K_, N_, device="cuda")).bfloat16()}]
    fn = (lambda s: F.linear(s["a"], s["w"])) if trans_b else (lambda s: s["a"] @ s["w"])
    fns = rotating(fn, sets[0], ("a",), 2 * (M_ * K_ + M_ * N_))
    t = time_kernel(fns)
    return round(2.0 * M_ * N_ * K_ / t / 1e12, 1)


# --------------------------------------------------------------------------- CPU baseline
def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (port of the reference's pure-PyTorch FastVim path incl. selective_scan_ref)
    running the SAME workload -- FastVim-T 224x224 fwd+bwd, fp32 -- on a bounded sample."""
    from oracle import fastvim_forward_oracle, make_state_dict, selective_scan_ref_port
    # intra-op threading of small CPU tensors stops scaling (and then regresses) well before the
    # 256 hardware threads of the GPU host: use at most 32, and report that number
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.requires_grad_() for k, v in make_state_dict(seed=0, embed_dim=192, depth=24).items()}
    bs = 2
    x = torch.randn(bs, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    t = torch.softmax(torch.randn(bs, 1000, generator=torch.Generator().manual_seed(1)), -1)

    def step():
        logits = fastvim_forward_oracle(sd, x, compute_dtype=torch.float32)
        loss = torch.sum(-t * F.log_softmax(logits, -1), -1).mean()
        loss.backward()

    tw = time.perf_counter()
    step()                                   # warm (allocator, thread pools)
    warm_s = time.perf_counter() - tw
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el + el / n > seconds_budget - warm_s or n >= 24:
            break
    ips = n * bs / el
    # the scan op alone at the benchmark shape (B, d_in, Lc, N) = (128, 384, 14, 16)
    g = torch.Generator().manual_seed(0)
    u, dl = torch.randn(128, 384, 14, generator=g), 0.5 * torch.rand(128, 384, 14, generator=g)
    A = -0.5 * torch.rand(384, 16, generator=g)
    Bm, Cm = torch.randn(128, 16, 14, generator=g), torch.randn(128, 16, 14, generator=g)
    db = 0.5 * torch.rand(384, generator=g)
    selective_scan_ref_port(u, dl, A, Bm, Cm, None, None, db, True)
    t1 = time.perf_counter()
    for _ in range(3):
        selective_scan_ref_port(u, dl, A, Bm, Cm, None, None, db, True)
    scan_ms = (time.perf_counter() - t1) / 3 * 1e3
    return {"value": round(ips, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"FastVim-T 224x224 fp32 fwd+bwd, bs={bs}, {n} steps through the CPU oracle "
                      f"(pure-PyTorch port incl. selective_scan_ref); scan op alone at (128,384,14,16): "
                      f"{scan_ms:.1f} ms/call",
            "scan_ref_ms_128x384x14x16": round(scan_ms, 2)}


# --------------------------------------------------------------------------- scan op at the config shapes
SCAN_SHAPES = {      # BASELINE configs -> (B, d_in, Lc, N) of one selective_scan_fn call  
    "cfg2_FastVimT_224_bs128": (128, 384, 14, 16),
    "cfg3_FastVimB_224_bs128": (128, 1536, 14, 16),
    "cfg4_FastVimB_2048_bs8": (8, 1536, 128, 16),
    "cfg5_ChannelVimS_8ch_bs64": (64, 768, 112, 16),
}


def scan_op_table(cpu=True):
    """The reference-layout op `selective_scan_fn` (csrc/scan_bdl.hip: wave-level associative scan) at every
    config's (B, d_in, Lc, N): device time fwd and fwd+bwd, HBM GB/s on the ALGORITHMIC bytes of SURVEY.md
    section 8d, and the CPU oracle's port of selective_scan_ref on the same inputs beside it."""
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    out = {}
    for name, (B, D, Lc, N) in SCAN_SHAPES.items():
        g = torch.Generator().manual_seed(0)
        u, dl = torch.randn(B, D, Lc, generator=g), 0.5 * torch.rand(B, D, Lc, generator=g)
        A = -0.5 * torch.rand(D, N, generator=g)
        Bm, Cm = torch.randn(B, N, Lc, generator=g), torch.randn(B, N, Lc, generator=g)
        db = 0.5 * torch.rand(D, generator=g)
        e = 2
        bytes_f = e * (3 * B * D * Lc + 2 * B * N * Lc) + 4 * (D * N + D)
        bytes_b = e * (5 * B * D * Lc + 2 * B * N * Lc) + 4 * (2 * B * N * Lc + 2 * D * N + 2 * D)
        # HBM-cold: the calls cycle through enough copies of the inputs to push each out of the Infinity Cache
        nsets = min(48, max(3, -(-(IC_BYTES + (96 << 20)) // bytes_f) + 1))
        Ag, dbg = A.cuda().requires_grad_(), db.cuda().requires_grad_()
        qs, gos = [], []
        for _ in range(nsets):
            qs.append([t.cuda().bfloat16().requires_grad_() for t in (u, dl, Bm, Cm)])
            gos.append(torch.randn(B, D, Lc, device="cuda").bfloat16())

        def fwd(q):
            with torch.no_grad():
                return selective_scan_fn(q[0], q[1], Ag, q[2], q[3], None, None, dbg, True)

        def fwd_bwd(q, go):
            y = selective_scan_fn(q[0], q[1], Ag, q[2], q[3], None, None, dbg, True)
            return torch.autograd.grad(y, q + [Ag, dbg], go)

        tf = time_kernel([lambda q=q: fwd(q) for q in qs], iters=nsets)
        tfb = time_kernel([lambda q=q, go=go: fwd_bwd(q, go) for q, go in zip(qs, gos)], iters=nsets)
        tf_w = time_kernel(lambda: fwd(qs[0]), iters=10)
        row = {"shape_B_D_L_N": [B, D, Lc, N], "operand_sets": nsets, "fwd_us": round(tf * 1e6, 1),
               "fwd_us_warm": round(tf_w * 1e6, 1), "fwd_GBps": round(bytes_f / tf / 1e9, 1),
               "fwd_hbm_frac": round(bytes_f / tf / 1e9 / HBM_PEAK_GBS, 4), "fwd_bwd_us": round(tfb * 1e6, 1),
               "bwd_GBps": round(bytes_b / max(tfb - tf, 1e-9) / 1e9, 1), "algorithmic_MB_fwd": round(bytes_f / 1e6, 2),
               "algorithmic_MB_bwd": round(bytes_b / 1e6, 2)}
        if cpu:
            from oracle import selective_scan_ref_port
            selective_scan_ref_port(u, dl, A, Bm, Cm, None, None, db, True)
            t1 = time.perf_counter()
            selective_scan_ref_port(u, dl, A, Bm, Cm, None, None, db, True)
            row["cpu_ref_fwd_ms"] = round((time.perf_counter() - t1) * 1e3, 1)
        out[name] = row
    return out


def run_training_steps(model_name, img, batch, channels, dtype, steps, warmup, rank, world, dev, use_graph=True,
                       trace=False, buckets=3, comm_dtype=None, force_segmented=False):
    """Build the model + flat training state + fused optimizer, capture the whole step (fwd + loss + bwd + AdamW + EMA;
    the gradient exchange sits between graph replay and optimizer when world > 1) and time exactly ``steps`` steps
    after ``warmup`` untimed ones, bracketed by barrier + synchronize.  Returns (seconds, final loss, extras)."""
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    from fastvim_amd.losses import SoftTargetCrossEntropy

    torch.manual_seed(1234)                    # identical init on every rank (DDP broadcast equivalent)
    drop_path = {"T": 0.05, "S": 0.15, "B": 0.4, "C": 0.1, "V": 0.05, "M": 0.0, "MV": 0.0, "CV": 0.1}[model_name]   # imagenet_classification/config/FastVim*.yaml:15
    model = build_model(model_name, img, drop_path, channels).to(dev).train()
    gen = torch.Generator().manual_seed(100 + rank)
    in_ch = channels if model_name in ("C", "CV") else 3
    is_mae = model_name in ("M", "MV")
    x = torch.randn(batch, in_ch, img, img, generator=gen).to(dev)
    tgt = soft_targets(batch, 1000, gen, dev)
    flat = FlatTrainingState(model, comm_dtype=comm_dtype)      # flat fp32 params / grads + bf16 shadow weights
    no_decay = {n for n, p in model.named_parameters()
                if p.ndim <= 1 or n.endswith(".bias") or n in model.no_weight_decay() or getattr(p, "_no_weight_decay", False)}
    # one fused kernel: AdamW (the reference recipe's two param groups) + ModelEmaV2 lerp + bf16 shadow refresh
    # MAE pre-training recipe: lr = blr * batch / 256 with blr 1.5e-4, betas (0.9, 0.95) (mae/config/pretrain_FastVimB.yaml:20,
    # mae/mae_imagenet.py); the classification recipe's 1e-3 without warm-up diverges on it within ~10 steps
    lr, betas = (1.5e-4 * batch * world / 256, (0.9, 0.95)) if is_mae else (1e-3, (0.9, 0.999))
    opt = FlatAdamW(flat, model, lr=lr, betas=betas, weight_decay=0.05, no_decay=no_decay, ema_decay=0.9999)
    torch.manual_seed(5678 + rank)             # per-rank DropPath streams
    criterion = SoftTargetCrossEntropy()
    loss_seed = torch.ones((), device=dev, dtype=torch.float32)      # d loss / d loss, allocated once outside the graph

    def fwd_bwd():
        flat.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
            if is_mae:
                loss = model(x, mask_ratio=0.75)[0]      # norm-pix MSE on the 75 % removed patches (mae_imagenet.py SSLModule)
            else:
                logits = model(x)
        if not is_mae:
            loss = criterion(logits, tgt)      # SoftTargetCrossEntropy (supervised_imagenet.py:83), fused value + gradient
        loss.backward(gradient=loss_seed)      # (the seed autograd would otherwise fill inside the captured step: a library launch)
        flat.finish_backward()
        return loss.detach()

    seg = None
    if use_graph and (world > 1 or force_segmented) and model_name in ("T", "S", "B", "C") and buckets > 0:
        # N > 1: the step is a chain of graphs (forward | backward of K runs of blocks | optimizer) with the bucketed
        # gradient all-reduce launched between them, so that it overlaps the remaining backward (fastvim_amd/pipeline.py)
        from fastvim_amd.pipeline import SegmentedTrainStep
        seg = SegmentedTrainStep(model, flat, opt, criterion, x, tgt, n_segments=buckets,
                                 amp_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
        timing = {"on": False}

        def step():
            return seg.step(time_exposed=timing["on"])
    elif use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd_bwd()
                if world == 1:
                    opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # (a live process group's watchdog thread polls events: "thread_local" keeps that legal during capture, pipeline.py)
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if world > 1 else "global"):
            loss_buf = fwd_bwd()
            if world == 1:
                opt.step()

        def step():
            graph.replay()
            if world > 1:
                flat.allreduce_sum_()
                opt.step(grad_scale=1.0 / world)
            return loss_buf
    else:
        def step():
            l = fwd_bwd()
            flat.allreduce_sum_()
            opt.step(grad_scale=1.0 / world)
            return l

    for i in range(warmup):
        loss = step()
        if trace and rank == 0:
            print(f"warmup {i} loss {float(loss):.5f}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if seg is not None:
        timing["on"] = True
        seg.exchange.timing = world > 1          # per-bucket launch-to-completion events (fastvim_amd/ddp.py)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    loss_val = float(loss)
    n_params = sum(p.numel() for p in model.parameters())
    extras = {"params": n_params}
    if world > 1:
        ex = getattr(flat, "exchange", None)
        # what every rank ran on, gathered so that the first SCALE record is self-describing: device index, device name,
        # the collective library's version as torch reports it (torch.cuda.nccl.version() IS RCCL's on ROCm)
        try:
            ccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            ccl = None
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", 0)), "device_index": torch.cuda.current_device(),
                "device": torch.cuda.get_device_name(), "pid": os.getpid()}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        extras["ddp_ranks"] = ranks_info
        extras["ddp_lib"] = {"collective_library": "rccl" if (ccl and torch.version.hip) else dist.get_backend(), "version": ccl,
                             "torch": torch.__version__, "hip": torch.version.hip, "native": True}
        extras["ddp"] = {
            "backend": dist.get_backend(), "ranks": dist.get_world_size(), "overlapped": seg is not None,
            "buckets": len(ex.bounds) if ex is not None else 1,
            "bucket_MB": [round((b - a) * 4 / 1e6, 2) for a, b in ex.bounds] if ex is not None else None,
            "wire_dtype": ex.wire_names() if ex is not None else None,
            # per bucket, in the order backward completes them: launch -> completion of its all-reduce (events on the
            # exchange's own stream); only the wait after the LAST backward graph is exposed to the step
            "bucket_allreduce_ms": (seg.exchange.bucket_ms() if seg is not None else None),
            "allreduce_exposed_ms": None if seg is None else round(seg.exposed_ms(), 3),
            "launcher_env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "MASTER_ADDR", "MASTER_PORT",
                                                              "OMP_NUM_THREADS", "NCCL_DEBUG", "FASTVIM_BENCH_ONE_GPU")}}
    flat.close()
    return elapsed, loss_val, extras


OTHER_CONFIGS = (      # BASELINE configs[2..4] (per-GPU shape) + the paper's comparison point; model, img, batch, channels
    ("cfg3_FastVimB_224_bs128", "B", 224, 128, 3),
    ("cfg4_FastVimB_2048_bs8", "B", 2048, 8, 3),
    ("cfg5_FastChannelVimS16_8ch_224_bs64", "C", 224, 64, 8),
    ("f2_VimT_224_bs128_unpooled_baseline", "V", 224, 128, 3),
)


def other_configs_block(dtype, rank, dev, steps=5, warmup=2):
    """The other BASELINE configurations at their per-GPU shape on this one GPU, a few steps each, with the kernel table of
    the FastVim-B shapes (dominant kernel and its roofline fraction, MFMA fraction of in_proj / out_proj)."""
    import gc
    out = {}
    for key, mname, img, batch, ch in OTHER_CONFIGS:
        try:
            el, lv, ex = run_training_steps(mname, img, batch, ch, dtype, steps, warmup, rank, 1, dev)
        except Exception as e:      # a failing side configuration must not lose the headline line
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            continue
        row = {"ms_per_step": round(el / steps * 1e3, 3), "images_per_sec": round(batch * steps / el, 1), "steps": steps,
               "warmup": warmup, "final_loss": round(lv, 4), "finite": lv == lv, "params_M": round(ex["params"] / 1e6, 2)}
        if mname in ("T", "S", "B", "C"):
            gs = img // 16
            d = {"T": 192, "S": 384, "B": 768, "C": 384}[mname]
            adt = torch.bfloat16 if dtype == "bf16" else torch.float32
            # (channel model: tokens_per_patch = channels, Channel-First -- the cell-walking conv kernels, pooled length rows x channels)
            kt = kernel_table(batch, gs, gs, d, 24, adt, tpp=ch if mname == "C" else 1)
            try:
                add_floors(kt, adt)
            except Exception as e_:
                row["floor_error"] = f"{type(e_).__name__}: {e_}"[:200]
            dom = max(kt, key=lambda k: kt[k]["us_per_step"])
            row["dominant_kernel"] = {"kernel": dom, "avg_us": kt[dom]["us"], "us_per_step": kt[dom]["us_per_step"],
                                      "GBps": kt[dom]["GBps"], "hbm_frac": round(kt[dom]["GBps"] / HBM_PEAK_GBS, 4),
                                      **({"TFLOPs": kt[dom]["TFLOPs"], "mfma_frac": kt[dom]["mfma_frac"]} if "TFLOPs" in kt[dom] else {})}
            row["mfma_frac"] = {k[5:]: kt[k]["mfma_frac"] for k in kt if k.startswith("gemm_") and "mfma_frac" in kt[k]}
            row["kernel_us"] = {k: kt[k]["us"] for k in kt}
            row["kernel_floor_ratio"] = {k: kt[k]["floor_ratio"] for k in kt if "floor_ratio" in kt[k]}
            row["kernel_hbm_frac"] = {k: round(kt[k]["GBps"] / HBM_PEAK_GBS, 3) for k in kt}
            row["roofline_step"] = step_roofline(kt, el / steps * 1e3)
            if d >= 384:      # the vendor library on the same shapes, same box, HBM-cold: how good is 0.4-0.5 of peak here?
                Mt, d_in_ = batch * gs * gs * (ch if mname == "C" else 1), 2 * d
                try:
                    lib_tf = {"in_proj_fwd": library_gemm_ceiling(Mt, 2 * d_in_, d), "out_proj_fwd": library_gemm_ceiling(Mt, d, d_in_),
                              "out_proj_dgrad": library_gemm_ceiling(Mt, d_in_, d, trans_b=False),
                              "in_proj_dgrad": library_gemm_ceiling(Mt, d, 2 * d_in_, trans_b=False)}
                    row["hipblaslt_TFLOPs"] = lib_tf
                    row["build_TFLOPs"] = {k: kt["gemm_" + k]["TFLOPs"] for k in lib_tf}
                except Exception as e_:
                    row["hipblaslt_TFLOPs"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}
            del kt
        out[key] = row
        gc.collect()
        torch.cuda.empty_cache()
    return out


def inference_throughput(model_name, img, batch, dev, steps=5, warmup=2):
    """Forward-only images/s (eval mode, no_grad, bf16 autocast, HIP-graph replay): the quantity the reference's headline
    claim is about (README.md:15: FastVim vs Vim inference speed at 2048 x 2048)."""
    import gc
    torch.manual_seed(1234)
    model = build_model(model_name, img, 0.0).to(dev).eval()
    x = torch.randn(batch, 3, img, img, generator=torch.Generator().manual_seed(7)).to(dev)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(x)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            out = fwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = bool(torch.isfinite(out.float()).all())
    del g, model, x, out
    gc.collect()
    torch.cuda.empty_cache()
    return {"ms_per_batch": round(el / steps * 1e3, 3), "images_per_sec": round(batch * steps / el, 2), "finite": ok}


def baselines_block(dev, steps=3, warmup=1):
    """SURVEY row f2 for the other two task families: the pooled model against its un-pooled baseline at the shape the
    reference's configs train -- MAE pre-training (FastVim-B vs Vim-B encoder, 224 px, batch 128, mask 0.75:
    mae/config/pretrain_{FastVimB,VimB}.yaml) and JUMP-CP-shaped classification (FastChannelVim-S/16 vs ChannelVim-S/16,
    8 channels, 224 px, batch 64: cell_imaging/config/{FastChannelVimS,ChannelVimS}.yaml).  Training steps on the same
    kernels; the baselines scan every (kept) token instead of the pooled rows."""
    out = {}
    for key, fast, base, batch, ch in (("mae_pretrain_224px_bs128", "M", "MV", 128, 3),
                                       ("channel_8ch_224px_bs64", "C", "CV", 64, 8)):
        row = {}
        for tag, name in (("pooled", fast), ("unpooled_baseline", base)):
            try:
                el, lv, ex = run_training_steps(name, 224, batch, ch, "bf16", steps, warmup, 0, 1, dev)
                row[tag] = {"model": name, "ms_per_step": round(el / steps * 1e3, 2), "images_per_sec": round(batch * steps / el, 1),
                            "finite": lv == lv, "params_M": round(ex["params"] / 1e6, 2)}
            except Exception as e:
                row[tag] = {"model": name, "error": f"{type(e).__name__}: {e}"[:300]}
        if all("ms_per_step" in row[t] for t in ("pooled", "unpooled_baseline")):
            row["train_speedup_pct"] = round((row["unpooled_baseline"]["ms_per_step"] / row["pooled"]["ms_per_step"] - 1.0) * 100.0, 1)
        out[key] = row
    return out


def vim_vs_fastvim_block(dev, img=2048, batch=8):
    """the un-pooled Vim baseline against FastVim at the resolution of the paper's headline claim ("up to
    72.5 % speedup in inference speed ... on high resolution (2048x2048) images", /root/reference README.md:15).  Same
    kernels, same widths; Vim scans all L + 1 = 16 385 tokens per direction, FastVim the 128 pooled rows."""
    out = {"img": img, "batch": batch, "reference_claim": "up to 72.5 % speedup in inference speed at 2048 x 2048 (README.md:15)"}
    for size, fv_name, vim_name in (("T", "T", "V"),):
        try:
            a = inference_throughput(fv_name, img, batch, dev)
            b = inference_throughput(vim_name, img, batch, dev)
            out[f"FastVim-{size}"], out[f"Vim-{size}"] = a, b
            out[f"speedup_{size}_pct"] = round((a["images_per_sec"] / b["images_per_sec"] - 1.0) * 100.0, 1)
        except Exception as e:
            out[f"error_{size}"] = f"{type(e).__name__}: {e}"[:300]
    try:      # and the training step of the baseline at that resolution (fwd + bwd + AdamW + EMA)
        el, lv, ex = run_training_steps("V", img, batch, 3, "bf16", 3, 1, 0, 1, dev)
        out["Vim-T_train_ms_per_step"] = round(el / 3 * 1e3, 2)
        el, lv2, ex = run_training_steps("T", img, batch, 3, "bf16", 3, 1, 0, 1, dev)
        out["FastVim-T_train_ms_per_step"] = round(el / 3 * 1e3, 2)
        out["train_finite"] = lv == lv and lv2 == lv2
    except Exception as e:
        out["error_train"] = f"{type(e).__name__}: {e}"[:300]
    return out


# --------------------------------------------------------------------------- main
def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: run ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`` as a CHILD process
    (the reference's launcher equivalent: Lightning spawns one process per device, imagenet_classification/train.py:34-43)
    and relay its output.  Called before anything in this process has initialised HIP: the parent never owns a GPU context,
    and it never replaces itself (no exec) -- it waits for the child and returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this image
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, cwd=os.getcwd(), stdout=subprocess.PIPE, text=True)
    for line in child.stdout:          # rank 0's one JSON line (and nothing else: stderr goes straight through)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="T", choices=["T", "S", "B", "C", "V", "M", "MV", "CV"],
                    help="FastVim-T/S/B, C = FastChannelVim-S/16 (use --batch 64 for BASELINE configs[4]), V = Vim-T baseline")
    ap.add_argument("--channels", type=int, default=8, help="input channels of the channel model (--model C)")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (weak scaling)")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--kernels", action="store_true",
                    help="N > 1: time the kernel table as well (rank 0, after the timed region, the other ranks parked in a "
                         "barrier -- off by default there: the scaling run stays short and no rank idles next to a busy one)")
    ap.add_argument("--buckets", type=int, default=3,
                    help="N > 1: gradient buckets = backward graph segments the all-reduce overlaps with (0: one all-reduce after backward)")
    ap.add_argument("--segmented", action="store_true",
                    help="use the segmented (N > 1) step also on one GPU: measures what the chain of graphs costs by itself")
    ap.add_argument("--comm-dtype", default="fp32", choices=["auto", "fp32", "bf16"],
                    help="wire format of the gradient all-reduce: fp32 (default, the reference's DDP), bf16, or auto "
                         "(bf16 for buckets of >= 100 MB of fp32 gradient, else fp32) -- the compressed forms are opt-in")
    ap.add_argument("--no-combine-fusion", action="store_true",
                    help="A/B: combine as its own launch instead of inside the out_proj + add + norm launch (round 5)")
    ap.add_argument("--no-xproj-fold", action="store_true",
                    help="A/B: the x_proj adjoint as its own launch instead of inside the short scan backward (round 5)")
    ap.add_argument("--no-conv-dgrad", action="store_true",
                    help="A/B: the conv + pool adjoint as its own launch instead of inside the in_proj data gradient + norm adjoint (round 6)")
    ap.add_argument("--no-xproj-two-addends", action="store_true",
                    help="A/B: wide models add the x_proj adjoint's product to the fp32 d xc instead of handing it to the conv adjoint as a second bf16 addend (round 6)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the few-step runs of BASELINE configs 3, 4, 5 and the Vim-T baseline (default FastVim-T run only)")
    ap.add_argument("--vim-2048", action="store_true", help="only the Vim-vs-FastVim block at 2048 px (SURVEY row f2), as JSON")
    ap.add_argument("--no-scan-op", action="store_true",
                    help="skip timing the reference-layout op selective_scan_fn at every config's (B, d_in, Lc, N)")
    args = ap.parse_args()
    if args.gpus > 1 and not args.kernels:
        args.no_kernels = True
        args.no_other_configs = True
        args.no_scan_op = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU), BEFORE this process has
        # touched the GPU in any way -- a fresh child, never an exec -- and pass its JSON line and exit code on
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("FASTVIM_BENCH_HANG_DUMP"):      # debugging aid: python stacks of every thread after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["FASTVIM_BENCH_HANG_DUMP"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # test hook: FASTVIM_BENCH_ONE_GPU=1 puts every rank on GPU 0 and exchanges gradients through gloo, so the
    # multi-rank code path (graph replay -> all-reduce -> optimizer) can be exercised on a one-GPU box
    one_gpu = os.environ.get("FASTVIM_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    if os.environ.get("FASTVIM_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies (tests/test_pipeline_gpu.py)
        raise SystemExit(3)
    if args.vim_2048:
        print(json.dumps(vim_vs_fastvim_block(dev, args.img if args.img != 224 else 2048, args.batch if args.batch != 128 else 8)), flush=True)
        return
    if args.no_xproj_fold:
        import fastvim_amd.mamba_simple_faster as _msf
        _msf.XPROJ_IN_SCAN = False
    if args.no_combine_fusion:
        import fastvim_amd.mamba_simple_faster as _msf
        _msf.COMBINE_IN_OUT_PROJ = False
    if args.no_conv_dgrad:
        import fastvim_amd.mamba_simple_faster as _msf
        _msf.CONV_IN_DGRAD = False
    if args.no_xproj_two_addends:
        import fastvim_amd.mamba_simple_faster as _msf
        _msf.XPROJ_TWO_ADDENDS = False
    use_graph = not args.no_graph
    amp_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    trace = os.environ.get("FASTVIM_BENCH_TRACE") == "1"      # debugging aid: per-step loss (adds a sync per step)
    comm_dtype = {"auto": "auto", "fp32": None, "bf16": torch.bfloat16}[args.comm_dtype]
    elapsed, loss_val, extras = run_training_steps(args.model, args.img, args.batch, args.channels, args.dtype, args.steps,
                                                   args.warmup, rank, world, dev, use_graph=use_graph, trace=trace,
                                                   buckets=args.buckets, comm_dtype=comm_dtype,
                                                   force_segmented=args.segmented)
    if not (loss_val == loss_val):
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        gs = args.img // 16
        d = {"T": 192, "S": 384, "B": 768, "C": 384, "V": 192, "M": 768, "MV": 768, "CV": 384}[args.model]
        mname = (f"FastChannelVim-S/16 {args.channels}ch" if args.model == "C" else
                 "Vim-T (un-pooled baseline)" if args.model == "V" else
                 "FastVim-B MAE pre-training (mask 0.75, decoder 512x2)" if args.model == "M" else
                 "Vim-B MAE pre-training (un-pooled baseline, mask 0.75, decoder 512x2)" if args.model == "MV" else
                 f"ChannelVim-S/16 {args.channels}ch (un-pooled baseline)" if args.model == "CV" else f"FastVim-{args.model}")
        out = {
            "metric": "images/sec %s %dpx bs=%d/GPU fwd+bwd (+all-reduce +AdamW +EMA), whole job" % (mname, args.img, args.batch),
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "images_per_sec_per_gpu": round(value / world, 1),
            "config": {"workload": f"{mname} {args.img}x{args.img} bs={args.batch}/GPU {args.dtype} "
                                   f"training step, synthetic ImageNet tensors"
                                   + (" (BASELINE configs[1])" if (args.model, args.img, args.batch) == ("T", 224, 128) else "")
                                   + (" (BASELINE configs[4], HCS off)" if (args.model, args.img, args.batch, args.channels) == ("C", 224, 64, 8) else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "hip_graph": use_graph, "optimizer_in_step": True, "final_loss": round(loss_val, 4),
                       "final_loss_hex": float(loss_val).hex()},
        }
        if "ddp" in extras:
            out["ddp"] = extras["ddp"]
            out["ddp"]["ranks_on"] = extras.get("ddp_ranks")
            out["ddp"]["library"] = extras.get("ddp_lib")
        if args.segmented:
            out["config"]["segmented_step"] = args.buckets
        if not args.no_kernels and args.model not in ("V", "M", "MV", "CV"):
            kt = kernel_table(args.batch, gs, gs, d, 24, amp_dtype, tpp=args.channels if args.model == "C" else 1)
            try:      # per row: what a plain copy of the row's own bytes costs on this box (floor_us) and us / floor_us
                out["floor_curve_copy_MB_us"] = add_floors(kt, amp_dtype)
            except Exception as e_:
                out["floor_curve_copy_MB_us"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}
            dom = max(kt, key=lambda k: kt[k]["us_per_step"])     # the kernel that costs the most time per step
            if (args.model, args.img, args.batch, args.dtype) == ("T", 224, 128, "bf16"):
                for k_, v_ in in_step_trace_us().items():         # the same kernel inside the step (committed rocprofv3 trace)
                    if k_ in kt:
                        kt[k_]["us_in_step_trace"] = v_
            out["kernels"] = kt
            traffic = None      # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/)
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_JSON)))["kernels"]
                if (args.model, args.img, args.batch, args.dtype) == ("T", 224, 128, "bf16") and dom in pm:
                    traffic = pm[dom]["traffic_bytes"]
            except (OSError, KeyError, ValueError):
                pass
            out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": kt[dom]["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(kt[dom]["GBps"] / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "avg_us": kt[dom]["us"], "floor_us": kt[dom].get("floor_us"),
                               "floor_ratio": kt[dom].get("floor_ratio"),
                               "algorithmic_bytes": int(kt[dom]["algorithmic_MB"] * 1e6),
                               # avg_us is HBM-cold (operand sets rotated past the 256 MB Infinity Cache, as the step sees
                               # the kernel); the cache-resident repeat of one operand set is kept beside it
                               "timing": "hbm_cold_rotating_operands", "operand_sets": kt[dom].get("operand_sets"),
                               "avg_us_in_step_trace": kt[dom].get("us_in_step_trace"),
                               "avg_us_warm": kt[dom].get("us_warm"),
                               "frac_warm": (round(kt[dom]["algorithmic_MB"] * 1e6 / (kt[dom]["us_warm"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                             if kt[dom].get("us_warm") else None)}
            out["roofline"]["committed_profile"] = profile_stamp()
            out["roofline_step"] = step_roofline(kt, ms)
            if (args.model, args.img, args.batch, args.dtype) == ("T", 224, 128, "bf16"):
                # kernel_rows_us sums the HBM-COLD stand-alone timings of the table (cold inputs overstate kernels whose input
                # the previous launch has just written, the scans above all); the in-step sum comes from the committed trace
                # and includes the launches the table does not time (reductions, patch unfold, optimizer, library nodes)
                out["roofline_step"]["kernel_rows_us_source"] = "hbm_cold_stand_alone_timings_of_this_run"
                stb = step_trace_breakdown()
                if stb is not None:
                    out["roofline_step"]["kernel_us_per_step_in_committed_trace"] = stb["kernel_us_per_step"]
                    out["step_trace"] = stb
            try:      # plain elementwise kernels on cold tensors of the same size: what a 10-15 us launch can reach at all
                out["roofline"]["elementwise_floor_same_size_cold"] = stream_floor(args.batch, gs * gs * (args.channels if args.model == "C" else 1), 2 * d, amp_dtype)
            except Exception as e_:
                out["roofline"]["elementwise_floor_same_size_cold"] = {"error": f"{type(e_).__name__}: {e_}"[:200]}
            if dom.startswith("scan"):
                # the pooled scan moves 1/cols of a full-length tensor: it is bound by VALU issue, not by HBM
                # (profiles/r01_pmc_scan_bwd.json: ~85 % of the SIMD issue slots busy at 4 waves/SIMD); the
                # HBM-bound and MFMA-bound kernels of the step are reported next to it
                out["roofline"]["note"] = "VALU-issue bound (pooled tensors are 1/cols of full length); see roofline_others"
            hbm_rows = [k for k in kt if not k.startswith(("scan", "gemm"))]
            gemm_rows = [k for k in kt if k.startswith("gemm")]
            hb = max(hbm_rows, key=lambda k: kt[k]["us_per_step"])
            gm = max(gemm_rows, key=lambda k: kt[k]["us_per_step"])
            out["roofline_others"] = {
                "largest_hbm_bound_kernel": {"kernel": hb, "bound": "hbm", "achieved": kt[hb]["GBps"], "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": round(kt[hb]["GBps"] / HBM_PEAK_GBS, 4),
                                             "avg_us": kt[hb]["us"]},
                "largest_gemm": {"kernel": gm, "bound": "mfma", "achieved": kt[gm]["TFLOPs"], "peak": 2500.0,
                                 "unit": "TFLOP/s", "frac": kt[gm]["mfma_frac"], "hbm_GBps": kt[gm]["GBps"],
                                 "avg_us": kt[gm]["us"]}}
        if not args.no_cpu_baseline and world == 1 and args.model not in ("C", "V", "M", "MV", "CV"):
            out["cpu_baseline"] = cpu_baseline()
        if not args.no_scan_op and not args.no_kernels and world == 1 and args.model == "T":
            out["scan_op"] = scan_op_table(cpu=not args.no_cpu_baseline)
        if (not args.no_other_configs and world == 1 and use_graph
                and (args.model, args.img, args.batch, args.dtype) == ("T", 224, 128, "bf16")):
            out["other_configs"] = other_configs_block(args.dtype, rank, dev)
            out["other_configs"]["f2_vim_vs_fastvim_2048px"] = vim_vs_fastvim_block(dev)
            out["other_configs"]["f2_baselines_mae_and_channel"] = baselines_block(dev)
            # forward-only (inference) throughput at 224 px, batch 128 -- eval mode, no_grad, bf16 autocast, HIP-graph replay
            inf = {}
            for key, mname in (("FastVim-T", "T"), ("FastVim-B", "B"), ("Vim-T", "V")):
                try:
                    inf[key] = inference_throughput(mname, 224, 128, dev, steps=10, warmup=3)
                except Exception as e:
                    inf[key] = {"error": f"{type(e).__name__}: {e}"[:200]}
            out["other_configs"]["inference_224px_bs128"] = inf
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
