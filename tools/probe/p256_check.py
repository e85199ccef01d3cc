"""Phased 256 x 256 forward GEMM (FASTVIM_GEMM_P256, tuning build) against the per-tile kernels: bitwise equality of C over
several shapes and repeated launches (race screen), then timing.  usage: python tools/probe/p256_check.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
SHAPES = [("nt", 25088, 3072, 768), ("nt", 25088, 768, 1536), ("nt", 131072, 3072, 768), ("nt", 100352, 1536, 384),
          ("nt", 1000, 768, 128), ("nt", 25000, 1024, 1536), ("nt", 513, 2048, 192),
          ("nn", 131072, 768, 3072), ("nn", 131072, 1536, 768), ("nn", 100352, 768, 384), ("nn", 66000, 1024, 192),
          ("nn", 25088, 768, 3072), ("nn", 40000, 2048, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "gen":
    import torch
    from fastvim_amd.gemm import gemm_nn, gemm_nt
    from bench import time_kernel
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for (kind, M, N, K) in SHAPES:
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) if kind == "nt" else torch.randn(K, N, device="cuda", generator=g)).bfloat16()
        fn = gemm_nt if kind == "nt" else gemm_nn
        c0 = fn(a, w)
        same = True
        for _ in range(int(sys.argv[3])):
            same = same and torch.equal(fn(a, w), c0)
        t = time_kernel(lambda: fn(a, w), iters=10)
        out[(kind, M, N, K)] = c0.cpu()
        print(f"  {kind} {M}x{N}x{K}: repeat-equal {same}  {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
    torch.save(out, sys.argv[2])
else:
    import torch
    for tag, env in (("base", "0"), ("p256", "3")):
        print(tag, flush=True)
        e = dict(os.environ, FASTVIM_GEMM_P256=env)
        subprocess.run([sys.executable, __file__, "gen", f"/tmp/p256_{tag}.pt", "20"], env=e, check=True)
    a, b = torch.load("/tmp/p256_base.pt"), torch.load("/tmp/p256_p256.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(k, "bitwise" if torch.equal(a[k], b[k]) else f"DIFFERENT max abs {d}")
