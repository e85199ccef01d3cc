"""Forward / data-gradient GEMMs at the FastVim-T shapes, HBM-cold (FASTVIM_GEMM_STREAM* hooks select the kernel).  Rotates over buffer sets larger than the 256 MB
memory-side cache so that operands come from HBM as they do inside a training step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from fastvim_amd.gemm import gemm_nt, gemm_nn
from fastvim_amd import _lib as L
import ctypes
from bench import time_kernel
M = 25088
SETS = 8
def probe(tag, N, K, nn):
    As = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(SETS)]
    W = torch.randn(N, K, device="cuda").bfloat16()
    Wt = W.t().contiguous()
    Cs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(SETS)]
    i = [0]
    def fn():
        k = i[0] % SETS; i[0] += 1
        a, c = As[k], Cs[k]
        b = Wt if nn else W
        rc = L.lib().fv_gemm_bf16(L.ptr(a), L.ptr(b), L.ptr(c), None, L.i32(M), L.i32(N), L.i32(K), ctypes.c_long(K),
                                  ctypes.c_long(b.stride(0)), ctypes.c_long(N), L.i32(0), L.i32(1 if nn else 0), L.i32(0),
                                  L.i32(1), L.stream_of(a))
        L.check(rc, "gemm")
    fn()
    k = (i[0] - 1) % SETS
    ref = As[k].float() @ W.float().t()
    err = (Cs[k].float() - ref).abs().max().item() / ref.abs().max().item()
    t = time_kernel(fn, iters=24)
    by = 2.0 * (M * K + N * K + M * N)
    print(f"dbg={os.environ.get('FASTVIM_GEMM_DBG','0')} stream={os.environ.get('FASTVIM_GEMM_STREAM','0')} wg={os.environ.get('FASTVIM_GEMM_STREAM_WG','-')} bal={os.environ.get('FASTVIM_GEMM_STREAM_BAL','1')} err={err:.1e} {tag:16s} N{N} K{K} {'NN' if nn else 'NT'} {t*1e6:7.1f} us {by/t/1e9:7.0f} GB/s", flush=True)
for tag, N, K, nn in [("in_proj fwd", 768, 192, False), ("out_proj fwd", 192, 384, False), ("in_proj dgrad", 192, 768, True),
                      ("out_proj dgrad", 384, 192, True)]:
    probe(tag, N, K, nn)
