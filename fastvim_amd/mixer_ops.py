"""Thin typed wrappers over the fused-mixer C-ABI entry points (include/fastvim_hip.h).
No arithmetic happens here: allocate outputs with torch, pass pointers + sizes + stream."""
import ctypes

import torch

from . import _lib as L

f32 = ctypes.c_float


def _geo(rows, cols, transposed):
    # sequence position (i, j) -> memory token i*s_i + j*s_j  (see csrc/rowwalk.h)
    return (1, rows) if transposed else (cols, 1)


def conv_pool_fwd(xz, conv_w, conv_b, conv_w_b, conv_b_b, rows, cols, transposed, pool_max, scaling):
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    xc = torch.empty(2, B, rows, d_in, device=xz.device, dtype=xz.dtype)
    rc = L.lib().fv_mixer_conv_pool_fwd(
        L.ptr(xz), L.ptr(conv_w), L.ptr(conv_b), L.ptr(conv_w_b), L.ptr(conv_b_b), L.ptr(xc),
        L.i32(B), L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.i32(d_in), L.i32(conv_w.shape[-1]),
        L.i32(pool_max), f32(scaling), L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_conv_pool_fwd")
    return xc


def scan_fwd(xc, x_dbl, dt_w, dt_b, A_log, dt_w_b, dt_b_b, A_log_b):
    _, B, Lc, d_in = xc.shape
    R = dt_w.shape[1]
    N = A_log.shape[1]
    yc = torch.empty(2, B, Lc, d_in, device=xc.device, dtype=torch.float32)
    rc = L.lib().fv_mixer_scan_fwd(
        L.ptr(xc), L.ptr(x_dbl), L.ptr(dt_w), L.ptr(dt_b), L.ptr(A_log), L.ptr(dt_w_b), L.ptr(dt_b_b),
        L.ptr(A_log_b), L.ptr(yc), L.i32(B), L.i32(Lc), L.i32(d_in), L.i32(R), L.i32(N),
        L.i32(L.dtype_code(xc.dtype)), L.stream_of(xc))
    L.check(rc, "mixer_scan_fwd")
    return yc


def combine_fwd(xz, yc, conv_w, conv_b, conv_w_b, conv_b_b, D, D_b, ln_w, ln_b, eps, rows, cols, transposed):
    B, Ltok, two_d = xz.shape
    d_in = two_d // 2
    s_i, s_j = _geo(rows, cols, transposed)
    g = torch.empty(B, Ltok, d_in, device=xz.device, dtype=xz.dtype)
    if ln_w is not None:
        mean = torch.empty(B * Ltok, device=xz.device, dtype=torch.float32)
        rstd = torch.empty(B * Ltok, device=xz.device, dtype=torch.float32)
    else:
        mean = rstd = None
    rc = L.lib().fv_mixer_combine_fwd(
        L.ptr(xz), L.ptr(yc), L.ptr(conv_w), L.ptr(conv_b), L.ptr(conv_w_b), L.ptr(conv_b_b), L.ptr(D),
        L.ptr(D_b), L.ptr(ln_w), L.ptr(ln_b), f32(eps), L.ptr(g), L.ptr(mean), L.ptr(rstd), L.i32(B),
        L.i32(rows), L.i32(cols), L.i32(s_i), L.i32(s_j), L.i32(d_in), L.i32(conv_w.shape[-1]),
        L.i32(L.dtype_code(xz.dtype)), L.stream_of(xz))
    L.check(rc, "mixer_combine_fwd")
    return g, mean, rstd
