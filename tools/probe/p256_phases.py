"""Phased 256 x 256 GEMM at the FastVim-B shapes, HBM-cold, with parts of the kernel switched off (tuning build,
FASTVIM_GEMM_P256_DBG bits: 1 no C stores, 2 no counted waits, 4 no MFMAs, 8 no LDS reads, 16 no LDS-DMA loads): where a
256-row tile's time goes.  usage: python tools/probe/p256_phases.py M"""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L
from bench import time_kernel
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25088
SETS = 6
def probe(tag, N, K, nn):
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(SETS)]
    W = torch.randn(N, K, device="cuda").bfloat16()
    Wt = W.t().contiguous()
    Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(SETS)]
    def mk(k):
        a, c = As[k], Cs[k]
        b = Wt if nn else W
        def fn():
            rc = L.lib().fv_gemm_bf16(L.ptr(a), L.ptr(b), L.ptr(c), None, L.i32(M), L.i32(N), L.i32(K), ctypes.c_long(K),
                                      ctypes.c_long(b.stride(0)), ctypes.c_long(N), L.i32(0), L.i32(1 if nn else 0), L.i32(0),
                                      L.i32(1), L.stream_of(a))
            L.check(rc, "gemm")
        return fn
    t = time_kernel([mk(k) for k in range(SETS)], iters=24)
    fl = 2.0 * M * N * K
    tiles = -(-M // 256) * (N // 256)
    print(f"dbg={os.environ.get('FASTVIM_GEMM_P256_DBG','0'):>2s} {tag:15s} M{M} N{N} K{K} {'NN' if nn else 'NT'} {t*1e6:7.1f} us {fl/t/1e12:6.0f} TFLOP/s  "
          f"{tiles} tiles = {tiles/256:.2f} rounds, {t*1e6/(-(-tiles//256)):.1f} us per round, {K//64} K tiles", flush=True)
for tag, N, K, nn in [("in_proj fwd", 3072, 768, False), ("out_proj dgrad", 1536, 768, True), ("out_proj fwd", 768, 1536, False),
                      ("in_proj dgrad", 768, 3072, True)]:
    probe(tag, N, K, nn)
