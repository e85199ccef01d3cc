"""Where a workgroup of the short pooled-scan backward spends a batch element (tuning build: SC_STAMP in csrc/scan_cl.hip).
HBM-cold: launches rotate through operand sets, the stamps of the LAST launch are read.
usage: python -m fastvim_amd.build --tuning && python tools/probe/scan_stamps.py [d_model] [xproj]   (192: FastVim-T, 768: -B;
xproj: the kernel with the x_proj adjoint folded in, round 5)"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import _lib as L, mixer_ops as M
d = int(sys.argv[1]) if len(sys.argv) > 1 else 192
XPJ = len(sys.argv) > 2 and sys.argv[2] == "xproj"
B, Lc, d_in, Rk, N, dev = 128, 14, 2 * d, -(-d // 16), 16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
Wdt = rn(d_in, Rk) * Rk ** -0.5
bdt = torch.full((d_in,), -4.0, device=dev)
A_log = torch.log(torch.arange(1, N + 1, device=dev, dtype=torch.float32)).repeat(d_in, 1).contiguous()
SETS = 24
sets = [dict(xc=rn(2, B, Lc, d_in).bfloat16(), x_dbl=rn(2, B * Lc, Rk + 2 * N).bfloat16(), dyc=rn(B, Lc, d_in)) for _ in range(SETS)]
lib = L.lib()
nbb_rows = lib.fv_mixer_scan_bwd_partials(L.i32(B), L.i32(Lc), L.i32(Rk))       # B / NBB
NBB = B // nbb_rows
nwg = (d_in // 192) * nbb_rows * 2
stamps = torch.zeros(nwg * NBB, 12, device=dev, dtype=torch.int64)
Wx = rn(2, Rk + 2 * N, d_in) * d_in ** -0.5
lib._cdll.fv_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
real_reduce = M.reduce_partials
M.reduce_partials = lambda part, n, out=None, **kw: out if out is not None else part[0]
keep = []
def launch(s):
    if XPJ:
        keep.append(M.scan_bwd_xproj(s["xc"], s["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, s["dyc"], Wx[0], Wx[1]))
    else:
        keep.append(M.scan_bwd(s["xc"], s["x_dbl"], Wdt, bdt, A_log, Wdt, bdt, A_log, s["dyc"], keep_chunks=True))
for i in range(SETS):
    launch(sets[i])
torch.cuda.synchronize(); keep.clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(SETS - 1):
    launch(sets[i])
e0.record(); launch(sets[SETS - 1]); e1.record()
torch.cuda.synchronize()
lib._cdll.fv_debug_set_stamps(ctypes.c_void_p(0))
M.reduce_partials = real_reduce
us = e0.elapsed_time(e1) * 1e3
t = stamps.cpu().double().view(nwg, NBB, 12)
if not XPJ:
    t[:, :, 8:11] = t[:, :, 11:12]
ok = (t != 0).all(-1).all(-1)
t = t[ok]
names = ["loop top -> staged + barrier", "delta on the matrix cores, softplus, table", "forward recurrence", "adjoint sweep",
         "dt_proj adjoint (matrix cores), d u stores", "barrier", "12-wave sums (+ d x_dbl partial stores)",
         "x_proj: barrier + wait for the weight fragments", "x_proj: MFMAs", "x_proj: d u / d x_dbl stores", "end"]
LAST = 11
life = t[:, -1, LAST] - t[:, 0, 0]
print(f"d_model {d}: d_inner {d_in}, dt_rank {Rk}, {nwg} workgroups x {NBB} batch elements; launch {us:.1f} us (events); "
      f"{int(ok.sum())} workgroups fully stamped; workgroup life median {life.median().item():.0f} ticks (100 MHz s_memtime: /100 = us)")
per = t[:, :, 1:] - t[:, :, :-1]                       # (wg, element, 7 phases)
gap = t[:, 1:, 0] - t[:, :-1, LAST] if NBB > 1 else None
tot = (t[:, :, LAST] - t[:, :, 0]).median().item()
for k, nm in enumerate(names):
    m = per[:, :, k].median().item()
    print(f"  {nm:48s} {m:7.0f} ticks  {100 * m / tot:5.1f} %   (first element {per[:, 0, k].median().item():.0f}, last {per[:, -1, k].median().item():.0f})")
print(f"  element total (median) {tot:.0f} ticks; before the first element {(t[:, 0, 0] - t[:, 0, 0]).median().item():.0f}; "
      f"between elements {gap.median().item() if gap is not None else 0:.0f}")
