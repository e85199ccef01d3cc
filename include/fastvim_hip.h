/*
 * fastvim_hip.h -- C ABI of libfastvim_hip.so, the MI355X (gfx950) native kernel
 * library for the FastVim backbone hot path.
 *
 * Boundary rules:
 *   - extern "C", plain device pointers + sizes, no torch / C++ types;
 *   - every entry point enqueues work on `stream` (a hipStream_t) and returns
 *     immediately -- no host synchronisation, no allocation (graph-capture safe);
 *     the caller owns every buffer (in practice: the torch caching allocator);
 *   - return value: FV_OK (0) or a negative FV_ERR_* code; fv_last_error() gives
 *     the message (the Python host raises RuntimeError from it, mirroring the
 *     reference's TORCH_CHECK -> RuntimeError, selective_scan.cpp:235-305);
 *   - tensors are contiguous in the stated layout; `dtype` selects the storage
 *     type of activations (FV_F32 / FV_BF16 / FV_F16); all arithmetic is fp32.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the insitro/FastVim root).
 */
#ifndef FASTVIM_HIP_H
#define FASTVIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fv_stream_t; /* hipStream_t */

enum { FV_F32 = 0, FV_BF16 = 1, FV_F16 = 2 };
enum { FV_OK = 0, FV_ERR_INVALID = -1, FV_ERR_HIP = -2, FV_ERR_UNSUPPORTED = -3 };

const char* fv_last_error(void);
int fv_version(void);

/* ------------------------------------------------------------------------
 * Selective scan, reference layout (B, D, L) with L contiguous.
 * Replaces pybind `selective_scan_cuda.fwd`
 *   (mamba-1p1p1/csrc/selective_scan/selective_scan.cpp:226-336, 494-497) and
 * `selective_scan_cuda.bwd` (selective_scan.cpp:338-492).
 *
 *   u, delta, z, out : (batch, dim, seqlen)            storage `dtype`
 *   A                : (dim, dstate)                   fp32
 *   B, C             : (batch, n_groups, dstate, seqlen) storage `dtype` if *_variable
 *                      else (dim, dstate) fp32
 *   D, delta_bias    : (dim) fp32, nullable
 *   last_state       : (batch, dim, dstate) fp32, nullable
 * out = (scan(u, softplus?(delta + delta_bias), A, B, C) + D*u) * silu(z)
 * ---------------------------------------------------------------------- */
int fv_selective_scan_fwd(const void* u, const void* delta, const float* A, const void* B,
                          const void* C, const float* D, const void* z, const float* delta_bias,
                          void* out, float* last_state, int batch, int dim, int seqlen, int dstate,
                          int n_groups, int B_variable, int C_variable, int delta_softplus,
                          int dtype, fv_stream_t stream);

/* Workspace (bytes) fv_selective_scan_bwd needs in `workspace`. */
size_t fv_selective_scan_bwd_workspace(int batch, int dim, int seqlen, int dstate, int n_groups,
                                       int B_variable, int C_variable);

/* Gradients.  du, ddelta, dz: storage `dtype`.  dA (dim,dstate), dD, ddelta_bias (dim): fp32.
 * dB, dC: fp32, (batch, n_groups, dstate, seqlen) if variable else (dim, dstate).
 * All gradient buffers are overwritten (no accumulation).  Deterministic: no float atomics. */
int fv_selective_scan_bwd(const void* u, const void* delta, const float* A, const void* B,
                          const void* C, const float* D, const void* z, const float* delta_bias,
                          const void* dout, void* du, void* ddelta, float* dA, float* dB, float* dC,
                          float* dD, void* dz, float* ddelta_bias, void* workspace, int batch, int dim,
                          int seqlen, int dstate, int n_groups, int B_variable, int C_variable,
                          int delta_softplus, int dtype, fv_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused FastVim mixer "middle", channel-last (token-major) activations.
 * Together these replace the body of `Mamba.forward` between in_proj and out_proj
 *   (mamba-1p1p1/mamba_ssm/modules/mamba_simple_faster.py:270-444) and the fused autograd
 *   function `FastVim_MambaInnerFnNoOutProj_withoutZ`
 *   (mamba_ssm/ops/selective_scan_interface.py:452-776), which themselves call
 *   causal_conv1d_cuda.causal_conv1d_fwd/bwd (PyPI causal-conv1d 1.1.3.post1) and
 *   selective_scan_cuda.fwd/bwd.
 *
 * Layouts:  xz (batch, L, 2*d_inner)  [x | z] per token;  g, do (batch, L, d_inner);
 *           xc, yc, ... (2, batch, rows, d_inner)  [0] = forward direction, [1] = backward,
 *           both indexed by the pooled row in ORIGINAL order (no flips are materialised);
 *           x_dbl (2, batch*rows, dt_rank + 2*d_state) = [dt_low | B | C].
 * Token grid: the mixer's sequence position (i, j), i < rows, j < cols, is memory token
 *           i*tok_stride_row + j*tok_stride_col.  (cols, 1) = natural order; (1, rows) = the
 *           transposed grid odd layers see (models/fastvim.py:192-210) -- no copy is made.
 * Conv weights are (d_inner, d_conv) fp32 (= conv1d.weight viewed "d 1 w -> d w").
 * ---------------------------------------------------------------------- */
int fv_mixer_conv_pool_fwd(const void* xz, const float* conv_w, const float* conv_b,
                           const float* conv_w_b, const float* conv_b_b, void* xc, int batch, int rows,
                           int cols, int tok_stride_row, int tok_stride_col, int d_inner, int d_conv,
                           int pool_max, float scaling_factor, int dtype, fv_stream_t stream);

/* dt_proj + softplus + selective scan (A = -exp(A_log)) for both directions.  yc is fp32. */
int fv_mixer_scan_fwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                      const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                      const float* A_log_b, float* yc, int batch, int Lc, int d_inner, int dt_rank,
                      int d_state, int dtype, fv_stream_t stream);

/* g = LayerNorm(((yc_f + D*conv_f(x)) + (yc_b + D_b*conv_b(x))) / 2) * silu(z); ln_w == NULL
 * skips the norm (use_norm_after_ssm=False).  mean/rstd (batch*L) fp32 are saved for backward. */
int fv_mixer_combine_fwd(const void* xz, const float* yc, const float* conv_w, const float* conv_b,
                         const float* conv_w_b, const float* conv_b_b, const float* D, const float* D_b,
                         const float* ln_w, const float* ln_b, float ln_eps, void* g, float* mean,
                         float* rstd, int batch, int rows, int cols, int tok_stride_row,
                         int tok_stride_col, int d_inner, int d_conv, int dtype, fv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTVIM_HIP_H */
