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
/* ABI version of this header; fv_version() returns the one the library was built with -- a caller compiled against
 * another value must not call anything else.  History:
 *   1  rounds 1-3.
 *   2  round 4 changed fv_mixer_scan_bwd_segments / fv_mixer_scan_bwd_seg_partials (a d_inner argument was inserted)
 *      without bumping the number; round 5 bumps it for that break, removes the opt-in fv_mixer_mid_fwd(_ok) and
 *      fv_gemm_bf16_addnorm_rw(_ok), and adds fv_mixer_scan_bwd_xproj(_ok), fv_mixer_conv_pool_bwd2(_ok),
 *      fv_chunk_rows_bf16.
 *   3  round 6 adds fv_mixer_conv_pool_bwd_dgrad(_ok, _blocks), fv_transpose_bf16_batched, fv_gemm_bf16_tn_grouped_wide8
 *      (nothing removed or changed). */
#define FV_ABI_VERSION 3
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
 * Causal depthwise conv1d (+ SiLU), reference op layout (batch, dim, seqlen), seqlen contiguous.
 * Replaces `causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias, seq_idx=None, silu)` and
 * `causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias, dout, seq_idx=None, dx, silu) -> (dx, dw, dbias)`
 * of PyPI causal-conv1d 1.1.3.post1 (third-party, not vendored; pinned by the reference README.md:43 and
 * called at mamba_ssm/modules/mamba_simple_faster.py:274-285 and
 * mamba_ssm/ops/selective_scan_interface.py:496-498, 640-642, 751-753).
 *
 *   x, y, dy, dx : (batch, dim, seqlen) storage `dtype`;  weight (dim, width) fp32, width in 2..4;
 *   bias (dim) fp32, nullable;  silu != 0 applies SiLU to the output.
 *   y[b,d,l] = act(bias[d] + sum_k weight[d,k] * x[b,d,l-(width-1)+k])
 * Backward writes dx and per-(batch, dim) partials (batch, dim, 5) = [dw front-padded to 4 taps | dbias];
 * the caller sums them over batch with fv_reduce_partials (fixed order, deterministic).
 * ---------------------------------------------------------------------- */
int fv_causal_conv1d_fwd(const void* x, const float* weight, const float* bias, void* y, int batch, int dim,
                         int seqlen, int width, int silu, int dtype, fv_stream_t stream);
int fv_causal_conv1d_bwd(const void* x, const float* weight, const float* bias, const void* dy, void* dx,
                         float* partials, int batch, int dim, int seqlen, int width, int silu, int dtype,
                         fv_stream_t stream);

/* Expand + skip epilogue of the "compressed scan" fork: out[b,d,l] = yc[b,d,l/cf] + D[d]*u_full[b,d,l],
 * cf = seqlen / seqlen_compressed.  With fv_selective_scan_fwd on (u_compressed, delta, ...) this replaces
 * `faster_selective_scan_cuda.fwd(u, u_compressed, delta, A, B, C, D, z=None, delta_bias, softplus)`
 * (fastvim_kernel/mamba-1p1p1/csrc/selective_scan/selective_scan.cpp:216-360,
 *  selective_scan_fwd_kernel.cuh:68-299).  D nullable (then u_full may be null). */
int fv_scan_expand_skip_fwd(const void* yc, const void* u_full, const float* D, void* out, int batch, int dim,
                            int seqlen, int seqlen_compressed, int dtype, fv_stream_t stream);

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
 *           tokens_per_patch = t > 1 (channel-wise tokenization, Channel-First order,
 *           mamba_simple_channel_faster.py:242-256, 333-340): every grid cell holds t consecutive
 *           tokens, sequence position ((i*cols + j)*t + c), pooling groups (i, c), pooled tensors
 *           (2, batch, rows*t, d_inner), pooled index i*t + c; strides are in cells.
 * Conv weights are (d_inner, d_conv) fp32 (= conv1d.weight viewed "d 1 w -> d w").
 * ---------------------------------------------------------------------- */
/* Both depthwise convs + SiLU, the pooling over `cols` -> xc, and (skip != NULL) the D-weighted skip term
 * skip[b, token, :] = D*conv_f(x) + D_b*conv_b(x)  (batch, L, d_inner), memory token order, storage `dtype`
 * (mamba_simple_faster.py:356-358, 412-416): each conv+SiLU is evaluated once per forward pass.
 * pool_max != 0 (collapse_method="max", mamba_simple_faster.py:299-305): xc holds the row maxima and amax
 * (same shape and dtype as xc, nullable) receives the column of the first maximum, for the backward pass. */
int fv_mixer_conv_pool_fwd(const void* xz, const float* conv_w, const float* conv_b,
                           const float* conv_w_b, const float* conv_b_b, const float* D, const float* D_b,
                           void* xc, void* skip, void* amax, int batch, int rows, int cols, int tok_stride_row,
                           int tok_stride_col, int tokens_per_patch, int d_inner, int d_conv, int pool_max,
                           float scaling_factor, int dtype, fv_stream_t stream);

/* dt_proj + softplus + selective scan (A = -exp(A_log)) for both directions.  yc is fp32. */
int fv_mixer_scan_fwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                      const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                      const float* A_log_b, float* yc, int batch, int Lc, int d_inner, int dt_rank,
                      int d_state, int dtype, fv_stream_t stream);
/* Training form for long pooled lengths (Lc > 16, dt_rank <= 48): the forward launch also leaves the state entering
 * every 16-step chunk in `ckpt` (fv_mixer_scan_ckpt_floats() fp32; 0 = shape not covered, pass NULL), laid out
 * (2, batch, ceil(Lc/16), d_inner, d_state); fv_mixer_scan_bwd_ckpt(ckpt_given = 1) then skips its own forward sweep --
 * the states of the reference's forward kernel that its backward kernel re-derives (selective_scan_bwd_kernel.cuh:75-528
 * recomputes them chunk by chunk from `x`, the per-chunk running state saved by selective_scan_fwd_kernel.cuh:67-345). */
size_t fv_mixer_scan_ckpt_floats(int batch, int Lc, int d_inner, int d_state, int dt_rank);
int fv_mixer_scan_fwd_ckpt(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                           const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                           const float* A_log_b, float* yc, float* ckpt, int batch, int Lc, int d_inner, int dt_rank,
                           int d_state, int dtype, fv_stream_t stream);
/* Segment-parallel form for long sequences on few batch elements (the un-pooled Vim baseline at 2048 px: 16 385 steps,
 * batch 8; mamba_simple.py:210-255 -- the reference scans it with a parallel scan over L,
 * selective_scan_fwd_kernel.cuh:67-345): time is cut into fv_mixer_scan_fwd_segments() runs of whole 16-step chunks that
 * are scanned side by side -- states reached from zero, a serial combine over the segments, then the scan proper from the
 * true entry states (the recurrence is linear in the state).  seg_ws: fv_mixer_scan_fwd_seg_floats() fp32 of workspace
 * (0 = one segment: pass NULL, the call is fv_mixer_scan_fwd_ckpt).  yc / ckpt as there. */
int fv_mixer_scan_fwd_segments(int batch, int Lc, int d_inner, int dt_rank);
size_t fv_mixer_scan_fwd_seg_floats(int batch, int Lc, int d_inner, int d_state, int dt_rank);
int fv_mixer_scan_fwd_seg(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                          const float* A_log, const float* dt_w_b, const float* dt_bias_b, const float* A_log_b,
                          float* yc, float* ckpt, float* seg_ws, int batch, int Lc, int d_inner, int dt_rank,
                          int d_state, int dtype, fv_stream_t stream);

/* x_proj + dt_proj + softplus + selective scan in ONE launch for short pooled lengths (Lc <= 16, bf16, dt_rank <= 48:
 * the 224 / 256 px grids): x_dbl (2, batch*Lc, dt_rank + 2*d_state) = xc @ x_proj_w2[dir]^T is computed on the matrix
 * cores inside the scan kernel (mamba_simple_faster.py:321-327 + 328-354), written out in bf16 for the backward pass and
 * consumed from LDS.  x_proj_w2: (2, dt_rank + 2*d_state, d_inner) bf16, both directions.  fv_mixer_xproj_scan_fwd_ok
 * tells whether a shape is covered (else: fv_mixer_xproj_fwd + fv_mixer_scan_fwd). */
int fv_mixer_xproj_scan_fwd_ok(int Lc, int d_inner, int dt_rank, int dtype);
int fv_mixer_xproj_scan_fwd(const void* xc, const void* x_proj_w2, const float* dt_w, const float* dt_bias,
                            const float* A_log, const float* dt_w_b, const float* dt_bias_b, const float* A_log_b,
                            void* x_dbl, float* yc, int batch, int Lc, int d_inner, int dt_rank, int d_state,
                            int dtype, fv_stream_t stream);

/* g = LayerNorm((yc_f + yc_b + skip) / 2) * silu(z), the scan outputs expanded over `cols`; ln_w == NULL
 * skips the norm (use_norm_after_ssm=False).  mean/rstd (batch*L) fp32 are saved for backward, which
 * rebuilds the normalised value from skip, yc, mean, rstd. */
int fv_mixer_combine_fwd(const void* xz, const void* skip, const float* yc, const float* ln_w,
                         const float* ln_b, float ln_eps, void* g, float* mean, float* rstd, int batch,
                         int rows, int cols, int tok_stride_row, int tok_stride_col, int tokens_per_patch,
                         int d_inner, int dtype, fv_stream_t stream);

/* ---- backward of the fused mixer middle --------------------------------------------
 * Number of persistent blocks a row-walking backward kernel launches (which = 0: combine_bwd,
 * 1: conv_pool_bwd); their `partials` buffers are (blocks, 2*d_inner) and (blocks, 12*d_inner) fp32. */
int fv_mixer_bwd_blocks(int batch, int rows, int d_inner, int tokens_per_patch, int which);

/* Adjoint of the LayerNorm + gate of fv_mixer_combine_fwd; the normalised value is rebuilt from the saved
 * skip, yc, mean, rstd.  dg: gradient wrt g.
 * Writes dz into the z half of dxz (batch, L, 2*d_inner), d_o (batch, L, d_inner) = gradient wrt the
 * averaged pre-norm value, dyc (batch, rows, d_inner) fp32 = 0.5 * sum_j d_o (gradient wrt BOTH
 * directions' scan outputs), and per-block partials [d ln_w (d_inner) | d ln_b (d_inner)]. */
int fv_mixer_combine_bwd(const void* dg, const void* xz, const void* skip, const float* yc,
                         const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                         void* dxz, void* d_o, float* dyc, float* partials, int batch, int rows, int cols, int tok_stride_row,
                         int tok_stride_col, int tokens_per_patch, int d_inner, int dtype, fv_stream_t stream);

/* Adjoint of fv_mixer_scan_fwd with the dt_proj adjoint fused in (replaces
 * selective_scan_cuda.bwd + the einsums of selective_scan_interface.py:698-723).
 *   dyc    (batch, Lc, d_inner) fp32          in
 *   dxc    (2, batch, Lc, d_inner) fp32       out: gradient wrt xc through the scan input u
 *   dx_dbl (fv_mixer_scan_bwd_chunks, 2, batch*Lc, dt_rank+2*d_state) fp32 out: sum over chunks
 *          = gradient wrt x_dbl
 *   ckpt   fv_mixer_scan_bwd_ckpt_floats() fp32 scratch
 *   partials (batch, 2, d_inner*(d_state + dt_rank + 1)): per-batch partials, per direction the
 *          segments [dA_log (d_inner*d_state) | d dt_proj.weight (d_inner*dt_rank) | d dt_proj.bias (d_inner)]
 *          (sum over batch with fv_reduce_partials). */
int fv_mixer_scan_bwd_chunks(int d_inner, int Lc, int dt_rank);
/* the same for a given batch: long pooled lengths pick the channel-chunk width by how many workgroups the launch has
 * (fv_mixer_scan_bwd_chunks answers for a large batch); size dx_dbl with this one */
int fv_mixer_scan_bwd_chunks_b(int batch, int d_inner, int Lc, int dt_rank);
/* rows of the `partials` buffer of fv_mixer_scan_bwd: (rows, 2, d_inner*(d_state + dt_rank + 1)) fp32 */
int fv_mixer_scan_bwd_partials(int batch, int Lc, int dt_rank);
size_t fv_mixer_scan_bwd_ckpt_floats(int batch, int Lc, int d_inner, int d_state, int dt_rank);
int fv_mixer_scan_bwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                      const float* A_log, const float* dt_w_b, const float* dt_bias_b, const float* A_log_b,
                      const float* dyc, float* dxc, float* dx_dbl, float* ckpt, float* partials, int batch,
                      int Lc, int d_inner, int dt_rank, int d_state, int dtype, fv_stream_t stream);
/* Same with one output gradient per direction: dyc_per_direction != 0 -> dyc is (2, batch, Lc, d_inner).  The MAE
 * masked mixer needs it: its two branches gather different rows per token
 * (mamba_simple_masked_faster.py:281-283, 311-314), so their pooled gradients differ. */
int fv_mixer_scan_bwd_dir(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                          const float* A_log, const float* dt_w_b, const float* dt_bias_b, const float* A_log_b,
                          const float* dyc, int dyc_per_direction, float* dxc, float* dx_dbl, float* ckpt,
                          float* partials, int batch, int Lc, int d_inner, int dt_rank, int d_state, int dtype,
                          fv_stream_t stream);
/* Same; ckpt_given != 0: `ckpt` holds the checkpoints written by fv_mixer_scan_fwd_ckpt for the same inputs (read, not
 * scratch).  Pooled lengths above 16 take the chunked kernel (16-step chunks, states of a chunk recomputed into
 * registers from its checkpoint); without given checkpoints it derives them with a forward sweep of its own. */
int fv_mixer_scan_bwd_ckpt(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                           const float* A_log, const float* dt_w_b, const float* dt_bias_b, const float* A_log_b,
                           const float* dyc, int dyc_per_direction, float* dxc, float* dx_dbl, float* ckpt,
                           int ckpt_given, float* partials, int batch, int Lc, int d_inner, int dt_rank, int d_state,
                           int dtype, fv_stream_t stream);
/* Segment-parallel form of the same (long sequences on few batch elements, ckpt_given != 0 only: the un-pooled Vim baseline at
 * high resolution): the adjoint recurrence is linear in the adjoint state like the forward one in the state, so the
 * segments of fv_mixer_scan_fwd_seg are walked side by side here too -- adjoint states reached from zero, a serial combine
 * last segment to first, then the backward kernel proper per segment.  seg_ws: fv_mixer_scan_bwd_seg_floats() fp32 (0: one
 * segment, pass NULL); partials then has fv_mixer_scan_bwd_seg_partials() rows (one per batch element and segment). */
int fv_mixer_scan_bwd_segments(int batch, int Lc, int d_inner, int dt_rank);
size_t fv_mixer_scan_bwd_seg_floats(int batch, int Lc, int d_inner, int d_state, int dt_rank);
int fv_mixer_scan_bwd_seg_partials(int batch, int Lc, int d_inner, int dt_rank);
/* slices of dx_dbl (one per channel chunk) of the launch fv_mixer_scan_bwd_seg makes: with a segment workspace the
 * 64-channel workgroups of the segment-parallel form, without it fv_mixer_scan_bwd_chunks_b() */
int fv_mixer_scan_bwd_seg_chunks(int batch, int d_inner, int Lc, int dt_rank, int seg_ws_given);
int fv_mixer_scan_bwd_seg(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias, const float* A_log,
                          const float* dt_w_b, const float* dt_bias_b, const float* A_log_b, const float* dyc,
                          int dyc_per_direction, float* dxc, float* dx_dbl, float* ckpt, int ckpt_given, float* partials,
                          float* seg_ws, int batch, int Lc, int d_inner, int dt_rank, int d_state, int dtype,
                          fv_stream_t stream);

/* fv_mixer_scan_bwd for the short pooled lengths WITH the data half of the x_proj adjoint folded in (reference:
 * selective_scan_interface.py:679-696 + 726-734, `dx = dx_dbl @ x_proj.weight` added to the scan's d u).  Built for
 * Lc in {14, 16}, d_inner == 384 (two 192-channel chunks), dt_rank <= 12: ask fv_mixer_scan_bwd_xproj_ok.
 *   x_proj_w, x_proj_w_b (dt_rank + 2 * d_state, d_inner) fp32: the weights as stored
 *   x_proj_w2_bf16 (2, dt_rank + 2 * d_state, d_inner) bf16: the two, rounded -- required for bf16 storage, where the
 *        product is taken on the bf16 matrix cores from bf16(d x_dbl) and these (the operands of the reference's autocast
 *        backward); fp32 storage multiplies the fp32 weights exactly and ignores it (may be NULL)
 *   dxc  (2, batch, Lc, d_inner) fp32: d u through the scan + (this chunk's partial d x_dbl) @ Wx, own channels
 *   dxc2 (2, batch, Lc, d_inner) storage dtype: the same product for the OTHER chunk's channels; the total gradient of
 *        the pooled conv output is dxc + dxc2 (fv_mixer_conv_pool_bwd2 adds them)
 *   dx_dbl (2, 2, batch * Lc, W) fp32: per-chunk partial rows (fv_chunk_rows_bf16 sums them for the weight gradient)
 *   partials: fv_mixer_scan_bwd_partials() rows, as for fv_mixer_scan_bwd.  Deterministic. */
int fv_mixer_scan_bwd_xproj_ok(int batch, int Lc, int d_inner, int dt_rank, int dtype);
int fv_mixer_scan_bwd_xproj(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias, const float* A_log,
                            const float* dt_w_b, const float* dt_bias_b, const float* A_log_b, const float* dyc,
                            const float* x_proj_w, const float* x_proj_w_b, const void* x_proj_w2_bf16, float* dxc,
                            void* dxc2, float* dx_dbl, float* partials, int batch, int Lc, int d_inner, int dt_rank,
                            int d_state, int dtype, fv_stream_t stream);
/* outs[j] (rows, WP) bf16 = sum over nchunks of partials[j] (nchunks, rows, width) fp32, WP = width rounded up to 8, pad
 * columns zero; up to 64 jobs of one shape in one launch (host arrays of device pointers).  The rows
 * fv_mixer_xproj_bwd2 publishes, bit for bit. */
int fv_chunk_rows_bf16(const float* const* partials, void* const* outs, int njobs, int nchunks, long rows, int width,
                       fv_stream_t stream);

/* ---- MAE masked mixer: kept tokens <-> pooling rows --------------------------
 * Replaces compute_row_means_constantdivide (index_add_ over the kept tokens, divide by cols;
 * mamba_simple_masked_faster.py:376-416) and the torch.gather that expands the scan output back to the kept tokens
 * (:281-283, 311-314), plus their adjoints -- the adjoint of one is the other.  Deterministic (no atomics).
 *   idx (2, batch, n_tokens) int32: pooling row of token t, per direction; values outside [0, rows) contribute /
 *       receive nothing.
 * fv_rows_segment_sum: out[dir][b][r][:] = scale * sum_{t : idx[dir][b][t] == r} in[dir][b][t][:]
 *   in  (2, batch, n_tokens, d_inner), or (batch, n_tokens, d_inner) shared by both directions when
 *       in_per_direction == 0;   out (2, batch, rows, d_inner).
 * fv_rows_gather: out[dir][b][t][:] = scale * in[dir][b][idx[dir][b][t]][:]
 *   in  (2, batch, rows, d_inner);   out (2, batch, n_tokens, d_inner).
 * dtypes FV_F32 / FV_BF16 per tensor; d_inner a multiple of 4. */
int fv_rows_segment_sum(const void* in, int in_dtype, int in_per_direction, const int* idx, void* out, int out_dtype,
                        int batch, int n_tokens, int rows, int d_inner, float scale, fv_stream_t stream);
int fv_rows_gather(const void* in, int in_dtype, const int* idx, void* out, int out_dtype, int batch, int n_tokens,
                   int rows, int d_inner, float scale, fv_stream_t stream);

/* Adjoint of fv_mixer_conv_pool_fwd plus the D-skip path: consumes d_o and the total gradient
 * wrt the pooled conv output dxc (2, batch, rows, d_inner) fp32; writes dx into the x half of
 * dxz and per-block partials [d conv_w (d_inner*4) | d conv_w_b (d_inner*4) | d conv_b | d conv_b_b |
 * dD | dD_b] (12*d_inner floats).  pool_max != 0: `amax` is the argmax tensor the forward wrote; the
 * pooled gradient goes to that column only (autograd of `.max(dim).values`, mamba_simple_faster.py:299-305). */
int fv_mixer_conv_pool_bwd(const void* xz, const void* d_o, const float* dxc, const float* conv_w,
                           const float* conv_b, const float* conv_w_b, const float* conv_b_b, const float* D,
                           const float* D_b, const void* amax, void* dxz, float* partials, int batch, int rows,
                           int cols, int tok_stride_row, int tok_stride_col, int tokens_per_patch, int d_inner,
                           int d_conv, int pool_max, float scaling_factor, int dtype, fv_stream_t stream);

/* Same, with the pooled gradient given as two addends: dxc (fp32) + dxc2 (storage dtype, nullable) -- the form
 * fv_mixer_scan_bwd_xproj produces.  dxc2 is taken by the whole-row kernel only: mean pooling, tokens_per_patch 1,
 * 14 or 16 columns, d_inner a multiple of 128 (fv_mixer_conv_pool_bwd2_ok). */
int fv_mixer_conv_pool_bwd2_ok(int rows, int cols, int tokens_per_patch, int d_inner, int pool_max);
int fv_mixer_conv_pool_bwd2(const void* xz, const void* d_o, const float* dxc, const void* dxc2, const float* conv_w,
                            const float* conv_b, const float* conv_w_b, const float* conv_b_b, const float* D,
                            const float* D_b, const void* amax, void* dxz, float* partials, int batch, int rows,
                            int cols, int tok_stride_row, int tok_stride_col, int tokens_per_patch, int d_inner,
                            int d_conv, int pool_max, float scaling_factor, int dtype, fv_stream_t stream);

/* out[i] (+)= sum_{s < n_partials} partials[s*n + i], fixed order (deterministic); accumulate != 0
 * adds into `out` (gradient accumulation straight into a parameter's .grad). */
int fv_reduce_partials(const float* partials, float* out, int n_partials, size_t n, int accumulate,
                       fv_stream_t stream);
/* Up to 208 such reductions in one launch (host arrays of device pointers / sizes; at most 65 535 partials of fewer
 * than 2^32 elements each; 96 until round 6). */
int fv_reduce_partials_multi(const float* const* partials, float* const* outs, const int* n_partials,
                             const size_t* ns, int njobs, int accumulate, fv_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused (per-sample scale) + residual add + RMSNorm / LayerNorm over the last dim.
 * Replaces the Triton kernels behind rms_norm_fn / layer_norm_fn
 *   (mamba-1p1p1/mamba_ssm/ops/triton/layernorm.py:66-121 fwd, 210-304 bwd; host 124-191, 307-399).
 *   r = x * row_scale[row / rows_per_scale] + residual   (row_scale, residual nullable)
 *   residual_out = r (nullable);  y = norm(r) * weight (+ bias)
 * x, residual, y, residual_out: (M, N) with independent storage dtypes (FV_F32 / FV_BF16);
 * weight, bias: (N) fp32; mean (LayerNorm only), rstd: (M) fp32.  N % 4 == 0, N <= 2048.
 * ---------------------------------------------------------------------- */
int fv_add_norm_blocks(int M);   /* rows of the (blocks, N) partial_dw / partial_db buffers */
int fv_add_norm_fwd(const void* x, int x_dtype, const void* residual, int residual_dtype,
                    const float* weight, const float* bias, const float* row_scale, int rows_per_scale,
                    void* y, int y_dtype, void* residual_out, int residual_out_dtype, float* mean,
                    float* rstd, int M, int N, float eps, int is_rms_norm, fv_stream_t stream);
/* r: the saved residual_out.  dresidual_in = d r (nullable), dx = d r * row_scale (nullable);
 * partial_dw/partial_db: per-block partial sums, reduce with fv_reduce_partials. */
int fv_add_norm_bwd(const void* dy, int dy_dtype, const void* dresidual_out, int dresidual_out_dtype,
                    const void* r, int r_dtype, const float* weight, const float* mean, const float* rstd,
                    const float* row_scale, int rows_per_scale, void* dx, int dx_dtype, void* dresidual_in,
                    int dresidual_in_dtype, float* partial_dw, float* partial_db, int M, int N,
                    int is_rms_norm, fv_stream_t stream);

/* ------------------------------------------------------------------------
 * bf16 MFMA GEMM  C[M][N] = sum_k A(m,k) B(k,n) (+ bias[n]), fp32 accumulate.
 * Replaces the cuBLAS GEMMs behind in_proj / out_proj (mamba_simple_faster.py:189-193, 435-444),
 * the patch-embed Conv2d (models/fastvim.py:95, k == stride) and their autograd adjoints.
 *   a_k_slow = 0: A(m,k) = A[m*lda + k]      a_k_slow = 1: A(m,k) = A[k*lda + m]
 *   b_k_slow = 0: B(k,n) = B[n*ldb + k]      b_k_slow = 1: B(k,n) = B[k*ldb + n]
 *   (0,0) activations x weight^T;  (0,1) data gradient with the weight as stored;  (1,1) weight
 *   gradient X^T Y.  splits > 1 cuts K across grid.z and writes `splits` fp32 partials (stride
 *   M*ldc) for fv_reduce_partials.  C is bf16 (c_fp32 = 0) or fp32; bias (N) fp32 nullable.
 * ---------------------------------------------------------------------- */
int fv_gemm_bf16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                 long ldb, long ldc, int a_k_slow, int b_k_slow, int c_fp32, int splits, fv_stream_t stream);

/* fp32-MFMA GEMM (v_mfma_f32_16x16x4_f32), batched:  C_b[M][N] = sum_k A_b(m,k) B_b(k,n) (+ bias[n]),  b < batch.
 * The projections in the reference's DEFAULT precision (fp32, imagenet_classification/train.py:17) -- F.linear / matmul /
 * bmm behind in_proj, x_proj, out_proj, patch embed, head and their adjoints (mamba_simple_faster.py:189-193, 321-327,
 * 435-444; models/fastvim.py:95, 537) -- and every shape fv_gemm_bf16 does not take (odd extents, unaligned rows).
 * Operand layouts as fv_gemm_bf16 (a_k_slow / b_k_slow); operands FV_F32 or FV_BF16 (widened exactly), C FV_F32 or
 * FV_BF16, fp32 accumulate in K order; any M, N, K, leading dimensions and alignment.  Batch b uses A + b*strideA,
 * B + b*strideB, C + b*strideC (elements): both scan directions of x_proj in one launch, or the K slices of a deterministic
 * split-K weight gradient (C = fp32 partials for fv_reduce_partials). */
int fv_gemm_f32(const void* A, int a_dtype, const void* B, int b_dtype, void* C, int c_dtype, const float* bias, int M,
                int N, int K, long lda, long ldb, long ldc, int a_k_slow, int b_k_slow, int batch, long strideA,
                long strideB, long strideC, fv_stream_t stream);

/* out_proj fused with the NEXT block's residual add + RMSNorm (mamba_simple_faster.py:435-444 followed by
 * models/fastvim.py:168-190 / layernorm.py:66-121):
 *   h = bf16_round(A W^T);  r = residual + row_scale[row / rows_per_scale] * h;  residual_out = r (fp32);
 *   rstd[row] = rsqrt(mean_n r^2 + eps);  y = bf16(r * rstd * norm_weight)
 * -- bit for bit fv_gemm_bf16 (bf16 C) followed by fv_add_norm_fwd (is_rms_norm), without writing and re-reading h.
 * A (M, K) bf16, W (N, K) bf16, residual / residual_out (M, N) fp32, row_scale nullable.  Built for N == 192 and
 * K % 64 == 0 (a workgroup owns whole rows); returns FV_ERR_UNSUPPORTED otherwise and the caller runs the two calls. */
int fv_gemm_bf16_addnorm(const void* A, const void* W, const float* residual, const float* norm_weight,
                         const float* row_scale, int rows_per_scale, void* y, float* residual_out, float* rstd, int M,
                         int N, int K, long lda, long ldw, float eps, fv_stream_t stream);

/* in_proj data gradient fused with the SAME block's RMSNorm + residual-add adjoint (mamba_simple_faster.py:189-193
 * backward followed by ops/triton/layernorm.py:210-304): d y = bf16_round(A W) (A (M, K) bf16 = d xz, W (K, N) bf16 =
 * in_proj.weight as stored); then exactly fv_add_norm_bwd(is_rms_norm) on it: d residual_in = rstd (d y w - xhat mean(d y
 * w xhat)) + d residual_out (fp32), d x = d residual_in * row_scale (bf16), partial_dw (fv_gemm_bf16_dgrad_addnorm_blocks(M),
 * N): per-workgroup sums of d y * xhat (reduce with fv_reduce_partials).  The data gradient itself never reaches HBM.
 * N == 192, K % 64 == 0; FV_ERR_UNSUPPORTED otherwise. */
int fv_gemm_bf16_dgrad_addnorm_blocks(int M);
int fv_gemm_bf16_dgrad_addnorm_bwd(const void* A, const void* W, const float* dresidual_out, const float* r,
                                   const float* rstd, const float* norm_weight, const float* row_scale,
                                   int rows_per_scale, void* dx, float* dresidual_in, float* partial_dw, int M, int N,
                                   int K, long lda, long ldw, fv_stream_t stream);

/* The same with a second GEMM phase: C2 (M, N2) bf16 = d x @ W2, W2 (N, N2) bf16 row-major -- the PREVIOUS block's out_proj
 * data gradient (d g = d x @ out_proj.weight), computed from the d x tile while it is still in LDS; bit-identical to
 * fv_gemm_bf16(b_k_slow = 1) on d x.  W2 null: no second phase.  N2 % 128 == 0. */
int fv_gemm_bf16_dgrad_addnorm_bwd2(const void* A, const void* W, const float* dresidual_out, const float* r,
                                    const float* rstd, const float* norm_weight, const float* row_scale,
                                    int rows_per_scale, void* dx, float* dresidual_in, float* partial_dw, int M, int N,
                                    int K, long lda, long ldw, const void* W2, void* C2, int N2, long ldw2,
                                    fv_stream_t stream);

/* fv_mixer_combine_fwd + fv_gemm_bf16_addnorm in ONE launch (round 5): the gated activations g of a block are produced
 * tile by tile inside the out_proj GEMM that also does the NEXT block's DropPath scale + residual add + RMSNorm
 * (reference: mamba_simple_faster.py:356, 412-414, 434-444; models/fastvim.py:168-190).  g (batch, L, 384) bf16, mean /
 * rstd_ln (batch * L) are written for the backward pass exactly as fv_mixer_combine_fwd writes them; y / residual_out /
 * rstd exactly as fv_gemm_bf16_addnorm does from that g.  Built for bf16, d_inner 384, d_model 192, tokens_per_patch 1
 * (fv_mixer_combine_out_proj_addnorm_ok); W (192, ldw) bf16 row-major = out_proj.weight. */
int fv_mixer_combine_out_proj_addnorm_ok(int batch, int rows, int cols, int tokens_per_patch, int d_inner, int d_model,
                                         int dtype);
int fv_mixer_combine_out_proj_addnorm(const void* xz, const void* skip, const float* yc, const float* ln_w,
                                      const float* ln_b, float ln_eps, void* g, float* mean, float* rstd_ln, int batch,
                                      int rows, int cols, int tok_stride_row, int tok_stride_col, const void* W, long ldw,
                                      const float* residual, const float* norm_weight, const float* row_scale,
                                      int rows_per_scale, void* y, float* residual_out, float* rstd, float eps,
                                      fv_stream_t stream);

/* fv_mixer_conv_pool_bwd2 + fv_gemm_bf16_dgrad_addnorm_bwd2 in ONE launch (round 6): the backward mirror of
 * fv_mixer_combine_out_proj_addnorm.  A workgroup owns four pooling rows of one image: it runs the conv + pool adjoint on
 * them (reference: FastVim_MambaInnerFnNoOutProj_withoutZ.backward, selective_scan_interface.py:607-776 -- causal_conv1d_bwd
 * of both directions, the mean-pool adjoint, the D-skip adjoint), writes the x half of dxz (batch, L, 768) once -- the
 * in_proj weight gradient reads it -- and keeps it in LDS as the A operand of the in_proj data gradient
 * `dxz @ in_proj.weight` (mamba_simple_faster.py:189-193; the z half of dxz, written by fv_mixer_combine_bwd, is read
 * here), whose epilogue is the block's residual add + RMSNorm adjoint (models/fastvim.py:168-190) and whose second phase
 * is the previous block's out_proj data gradient C2 = dx @ W2 -- arguments from `W_in_t` on as in
 * fv_gemm_bf16_dgrad_addnorm_bwd2, except that the weight is given TRANSPOSED: W_in_t (192, ldwt) bf16 row-major =
 * in_proj.weight^T (fv_transpose_bf16_batched), so that a lane's MFMA fragment is 16 contiguous bytes.
 * conv_partials (fv_mixer_conv_pool_bwd_dgrad_blocks(batch, rows), 12 * 384) and partial_dw (same row count, 192) are
 * per-workgroup partial rows for fv_reduce_partials: [dw | dw_b | db | db_b | dD | dD_b] as fv_mixer_conv_pool_bwd writes
 * them, and the norm weight's gradient.  dx / dresidual_in / C2 and the x half of dxz are bit-identical to the two
 * launches' (tests/test_convpool_dgrad_gpu.py); the parameter-gradient sums group rows differently (fp32 rounding).
 * Built for bf16, d_inner 384, d_model 192, cols 14 or 16, tokens_per_patch 1, mean pooling
 * (fv_mixer_conv_pool_bwd_dgrad_ok). */
int fv_mixer_conv_pool_bwd_dgrad_ok(int batch, int rows, int cols, int tokens_per_patch, int d_inner, int d_model,
                                    int pool_max, int dtype);
int fv_mixer_conv_pool_bwd_dgrad_blocks(int batch, int rows);
int fv_mixer_conv_pool_bwd_dgrad(const void* xz, const void* dskip, const float* dxc, const void* dxc2,
                                 const float* conv_w, const float* conv_b, const float* conv_w_b, const float* conv_b_b,
                                 const float* D, const float* D_b, void* dxz, float* conv_partials, int batch, int rows,
                                 int cols, int tok_stride_row, int tok_stride_col, float scaling, const void* W_in_t,
                                 long ldwt, const float* dresidual_out, const float* r, const float* rstd,
                                 const float* norm_weight, const float* row_scale, int rows_per_scale, void* dx,
                                 float* dresidual_in, float* partial_dw, const void* W2, void* C2, int N2, long ldw2,
                                 fv_stream_t stream);

/* dsts[j] (cols, rows) bf16 = srcs[j] (rows, cols)^T for up to 64 equal-shape matrices in one launch: the transposed
 * bf16 shadows of in_proj.weight that fv_mixer_conv_pool_bwd_dgrad reads (refreshed once per optimizer step). */
int fv_transpose_bf16_batched(const void* const* srcs, void* const* dsts, int njobs, int rows, int cols,
                              fv_stream_t stream);

/* fv_gemm_bf16_addnorm with a second GEMM phase: C2 (M, N2) bf16 = y @ W2^T, W2 (N2, N) bf16 row-major -- the block's
 * in_proj (mamba_simple_faster.py:189-193) computed from the normalised tile while it is still in LDS; bit-identical to
 * fv_gemm_bf16 on y.  W2 null: no second phase.  N2 % 128 == 0. */
int fv_gemm_bf16_addnorm2(const void* A, const void* W, const float* residual, const float* norm_weight,
                          const float* row_scale, int rows_per_scale, void* y, float* residual_out, float* rstd, int M,
                          int N, int K, long lda, long ldw, float eps, const void* W2, void* C2, int N2, long ldw2,
                          fv_stream_t stream);

/* Several weight gradients in one launch (queued until the end of the backward pass): problem i is
 * x_i (Kd_i, M_i)^T @ y_i (Kd_i, N_i) -> parts_i (splits_i, M_i, N_i) fp32 partials (sum with fv_reduce_partials);
 * the same arithmetic, tiling and fixed split order as fv_gemm_bf16(a_k_slow = b_k_slow = 1, c_fp32 = 1).
 * splits_i == -1: one K slice ADDED in place to the (M_i, N_i) fp32 matrix parts_i points at (the gradient itself; no
 * partial, nothing to reduce) -- for launches whose outputs are all multiples of 256 x 256 and at least 512 x 512, Kd_i a
 * multiple of 64 and >= 128; an error otherwise. */
int fv_gemm_bf16_tn_grouped(const void* const* x, const void* const* y, float* const* parts, const int* Kd,
                            const int* M, const int* N, const int* splits, int count, fv_stream_t stream);
/* Same with explicit row strides (elements) of x and y -- rows may be padded (ldx >= M, ldy >= N, multiples of 8; pad
 * elements must be finite): M and N then need not be multiples of 8.  ldx / ldy null: dense rows. */
int fv_gemm_bf16_tn_grouped_ld(const void* const* x, const void* const* y, float* const* parts, const int* Kd,
                               const int* M, const int* N, const int* ldx, const int* ldy, const int* splits, int count,
                               fv_stream_t stream);

/* Tile class the grouped launch runs an (M, N) weight gradient over Kd tokens on when every problem of the launch agrees:
 * 4 = 256 x 192 tiles, 5 = 192 x 256 tiles (eight waves; the d_model-384 outputs from 50 000 tokens on), 0 = not taken.  The
 * host groups problems per launch and picks their split-K factors by the same answer. */
int fv_gemm_bf16_tn_grouped_wide8(int M, int N, int Kd);

/* x_proj of both directions, bf16: x_dbl (2, M, width) = xc (2, M, d_inner) @ x_proj_w2 (2, width, d_inner)^T, fp32
 * accumulate (mamba_simple_faster.py:321-327; the F.linear behind `self.x_proj` / `self.x_proj_b`).  M = batch*Lc,
 * width = dt_rank + 2*d_state <= 112, d_inner a multiple of 32. */
int fv_mixer_xproj_fwd(const void* xc, const void* x_proj_w2, void* x_dbl, int M, int d_inner, int width,
                       fv_stream_t stream);

/* Adjoint of x_proj for both directions, fused with the sum of the scan backward's chunk partials
 * (replaces the einsum/addmm chain of selective_scan_interface.py:698-734):
 *   dx_dbl = sum_c dx_dbl_partials[c];  dxc += dx_dbl @ W;  dW_partials[slice] = dx_dbl[slice]^T @ xc[slice].
 * dx_dbl_partials (nchunks, 2, M, width) fp32; xc (2, M, d_inner) storage dtype; W (width, d_inner) fp32;
 * dxc (2, M, d_inner) fp32 in/out; dW_partials (fv_mixer_xproj_bwd_slices(M), 2, width, d_inner) fp32.
 * Returns FV_ERR_UNSUPPORTED for a width that is not built (the host then uses library GEMMs). */
int fv_mixer_xproj_bwd_slices(int M);
int fv_mixer_xproj_bwd(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                       const float* x_proj_w_b, float* dxc, float* dW_partials, int M, int d_inner, int width,
                       int dtype, fv_stream_t stream);
/* Same; `dx_dbl_bf16` (2, M, WP) bf16 with WP = width rounded up to 8 (pad columns zero), nullable, receives the
 * summed dx_dbl rows, and `dW_partials` may then be null: the weight gradient dx_dbl^T xc is left to a (grouped)
 * GEMM over exactly those bf16 rows -- what the reference's autocast backward multiplies
 * (selective_scan_interface.py:726-729). */
int fv_mixer_xproj_bwd2(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                        const float* x_proj_w_b, float* dxc, float* dW_partials, void* dx_dbl_bf16, int M, int d_inner,
                        int width, int dtype, fv_stream_t stream);
/* The data half of fv_mixer_xproj_bwd2 (dW_partials == NULL; `dx_dbl_bf16` required) with the TRANSPOSED bf16 copy of both
 * weights, `x_proj_w_t_bf16` (2, d_inner, width): bf16(summed dx_dbl) @ bf16(weight) on the bf16 matrix cores, fp32
 * accumulate -- the operands of the reference's autocast backward (selective_scan_interface.py:726-734).  The product is
 * added to `dxc` (fp32, in place), or, with `dxc2_bf16` (2, M, d_inner) non-NULL, WRITTEN there in bf16 and `dxc` left alone
 * (the conv + pool adjoint then takes both: fv_mixer_conv_pool_bwd2).  Built where fv_mixer_xproj_bwd3_ok says so (bf16
 * storage, d_inner a multiple of 64 from 768 up, width 56 / 64 / 80 / 96 / 112); any other call, or a NULL transposed
 * weight, is fv_mixer_xproj_bwd2 with the fp32 weights (and refuses `dxc2_bf16`).  (Added in round 6; ABI version
 * unchanged: new symbols.) */
int fv_mixer_xproj_bwd3_ok(int M, int d_inner, int width, int dtype);
int fv_mixer_xproj_bwd3(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                        const float* x_proj_w_b, const void* x_proj_w_t_bf16, float* dxc, void* dxc2_bf16,
                        void* dx_dbl_bf16, int M, int d_inner, int width, int dtype, fv_stream_t stream);

/* Soft-target cross-entropy, value and gradient in one call (replaces timm.loss.SoftTargetCrossEntropy as the
 * reference trainer uses it, imagenet_classification/supervised_imagenet.py:83, 109-115, and its autograd):
 *   loss_rows[b] = sum_c -target[b][c] * log_softmax(logits[b])[c];  loss[0] = mean_b loss_rows[b];
 *   dlogits[b][c] = (softmax(logits[b])[c] * sum_c' target[b][c'] - target[b][c]) / batch      (fp32)
 * logits (batch, classes) fp32 or bf16, target fp32; classes <= 2048.  Deterministic (fixed summation order). */
int fv_soft_target_ce(const void* logits, int logits_dtype, const float* target, float* loss_rows, float* loss,
                      float* dlogits, int batch, int classes, fv_stream_t stream);

/* ------------------------------------------------------------------------
 * Small memory-bound ops around the backbone (csrc/glue.hip): each replaces a string of library elementwise /
 * reduce launches of the reference step.
 * ---------------------------------------------------------------------- */
/* The k == stride patch-embed Conv2d (models/fastvim.py:95) as a GEMM: its A operand, unfolded and cast in one pass.
 *   out[b][gi*gw + gj][(c*ph + pi)*pw + pj] = img[b][c][gi*ph + pi][gj*pw + pj]
 * img (batch, chans, height, width), out (batch, gh*gw, chans*ph*pw); fp32 or bf16 each; pw % 8 == 0. */
int fv_patch_unfold(const void* img, int img_dtype, void* out, int out_dtype, int batch, int chans, int height,
                    int width, int ph, int pw, fv_stream_t stream);
/* fv_gemm_bf16(a_k_slow = b_k_slow = 0) with fp32 C = bf16_round(A B^T) + table[(m mod period)][n]: the patch
 * projection with conv bias and position embedding added in the epilogue (models/fastvim.py:95-101, 500) -- the
 * value an autocast F.linear followed by the fp32 add returns.  table (period, N) fp32. */
int fv_gemm_bf16_rowbias(const void* A, const void* B, float* C, const float* table, int period, int M, int N, int K,
                         long lda, long ldb, long ldc, fv_stream_t stream);
/* final_pool_type == "mean" (models/fastvim.py:529-531): out[b][d] = mean_l x[b][l][d] (fp32 sum, fixed order) and its
 * adjoint dx[b][l][d] = g[b][d] / tokens.  x (batch, tokens, dim), dim % 4 == 0, one dtype throughout. */
int fv_mean_pool_fwd(const void* x, void* out, int batch, int tokens, int dim, int dtype, fv_stream_t stream);
int fv_mean_pool_bwd(const void* g, void* dx, int batch, int tokens, int dim, int dtype, fv_stream_t stream);
/* Stochastic depth (timm DropPath, models/fastvim.py:175-178): table (mods, batch) holds U[0,1) draws on entry and
 * floor(keep[m] + U) * inv_keep[m] on return -- one row of per-sample scales per DropPath module. */
int fv_droppath_table(float* table, const float* keep, const float* inv_keep, int mods, int batch, fv_stream_t stream);
/* y[i] = x[i] * scale[0] (device scalar), cast to y_dtype: the loss gradient handed to the head. */
int fv_scale_cast(const float* x, const float* scale, void* y, int y_dtype, size_t n, fv_stream_t stream);
/* out[c] (+)= sum_r x[r][c], x (rows, cols) fp32 or bf16, fixed order: a bias gradient. */
int fv_column_sum(const void* x, int dtype, float* out, int rows, int cols, int accumulate, fv_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused AdamW (decoupled weight decay, bias correction; torch.optim.AdamW semantics) over a flat fp32
 * parameter buffer, optionally updating an EMA copy (timm ModelEmaV2: ema = d*ema + (1-d)*p) and the
 * bf16 shadow weights in the same pass.  Replaces the optimizer + EMA + autocast casts of the
 * reference step (imagenet_classification/supervised_imagenet.py:134-147, 270-276).
 * decay_mask: one byte per element (1 = weight decay applies).  lr, step: device scalars (fp32);
 * step is incremented by the call.  grad_scale multiplies every gradient element as it is read: 1, or
 * 1 / world_size when `grads` holds the SUM over data-parallel ranks (the mean torch DDP produces,
 * imagenet_classification/train.py:34-43, without a pass of its own).  n % 4 == 0.
 * ---------------------------------------------------------------------- */
int fv_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema,
                  void* shadow_bf16, const uint8_t* decay_mask, const float* lr, float* step, float beta1,
                  float beta2, float eps, float weight_decay, float ema_decay, float grad_scale, size_t n,
                  fv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTVIM_HIP_H */
