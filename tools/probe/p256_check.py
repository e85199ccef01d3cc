"""Phased 256 x 256 forward GEMM (FASTVIM_GEMM_P256, tuning build) against the per-tile kernels: bitwise equality of C over
several shapes and repeated launches (race screen), then timing.  usage: python tools/probe/p256_check.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
SHAPES = [(25088, 3072, 768), (25088, 768, 1536), (25088, 1536, 384), (131072, 3072, 768), (100352, 1536, 384),
          (1000, 768, 128), (25000, 1024, 1536), (256, 256, 128), (513, 2048, 192)]
if len(sys.argv) > 1 and sys.argv[1] == "gen":
    import torch
    from fastvim_amd.gemm import gemm_nt
    from bench import time_kernel
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for (M, N, K) in SHAPES:
        a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
        c0 = gemm_nt(a, w)
        same = True
        for _ in range(int(sys.argv[3])):
            same = same and torch.equal(gemm_nt(a, w), c0)
        t = time_kernel(lambda: gemm_nt(a, w), iters=10)
        out[(M, N, K)] = c0.cpu()
        print(f"  {M}x{N}x{K}: repeat-equal {same}  {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TFLOP/s", flush=True)
    torch.save(out, sys.argv[2])
else:
    import torch
    for tag, env in (("base", "0"), ("p256", "256")):
        print(tag, flush=True)
        e = dict(os.environ, FASTVIM_GEMM_P256=env)
        subprocess.run([sys.executable, __file__, "gen", f"/tmp/p256_{tag}.pt", "20"], env=e, check=True)
    a, b = torch.load("/tmp/p256_base.pt"), torch.load("/tmp/p256_p256.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(k, "bitwise" if torch.equal(a[k], b[k]) else f"DIFFERENT max abs {d}")
