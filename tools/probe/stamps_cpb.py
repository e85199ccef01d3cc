"""Per-wave timeline of conv_pool_bwd (library built with -DFV_DBG_STAMPS on convpool_bwd_row.hip)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from fastvim_amd import mixer_ops as M, _lib as L
B, rows, cols, d = 128, 14, 14, 192
dtype, dev = torch.bfloat16, "cuda"
d_in, Ltok = 2 * d, rows * cols
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s, dt=dtype: torch.randn(*s, device=dev, generator=g).to(dt)
xz = rn(B, Ltok, 2 * d_in)
cw, cwb = rn(d_in, 4, dt=torch.float32) * 0.5, rn(d_in, 4, dt=torch.float32) * 0.5
cb, cbb = rn(d_in, dt=torch.float32) * 0.1, rn(d_in, dt=torch.float32) * 0.1
D, Db = torch.ones(d_in, device=dev), torch.ones(d_in, device=dev)
d_o = rn(B, Ltok, d_in); dxc = rn(2, B, rows, d_in, dt=torch.float32); dxz = torch.empty_like(xz)
junk = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
for _ in range(3):
    if cold: junk.zero_()           # flush the caches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    M.conv_pool_bwd(xz, d_o, dxc, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0)
    e1.record()
    torch.cuda.synchronize()
    print("event ms (kernel + partial reduction)", e0.elapsed_time(e1))
from bench import time_kernel
print("time_kernel us:", time_kernel(lambda: M.conv_pool_bwd(xz, d_o, dxc, cw, cb, cwb, cbb, D, Db, dxz, rows, cols, False, 0, 1.0)) * 1e6)
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib = ctypes.CDLL(L.LIB_PATH)
rc = lib.fv_dbg_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0, rc
s = buf.reshape(4096, 8)[:3072].astype(np.int64)
two = s[:, 3] > 0
d = (s - s[:, :1]) * 0.01      # s_memtime ticks (100 MHz) relative to the wave's own start -> us
names = ["start", "row0 begin", "row0 end", "row1 end", "flushed"]
for i, n in enumerate(names):
    v = d[:, i] if i != 3 else d[two, i]
    print(f"{n:12s} min {v.min():6.2f} p10 {np.percentile(v,10):6.2f} med {np.median(v):6.2f} p90 {np.percentile(v,90):6.2f} max {v.max():6.2f}")
print("row0 duration, 1-row waves:", np.median(d[~two, 2] - d[~two, 1]), " 2-row waves:", np.median(d[two, 2] - d[two, 1]), " row1:", np.median(d[two, 3] - d[two, 2]))
print("waves with 2 rows:", two.sum(), "of", len(s))
