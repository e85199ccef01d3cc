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

#ifdef __cplusplus
}
#endif
#endif /* FASTVIM_HIP_H */
