// hipcc-flags: -fgpu-flush-denormals-to-zero
// Causal depthwise conv1d (+ optional SiLU) in the reference op layout (batch, dim, seqlen), seqlen
// contiguous.  Replaces causal_conv1d_cuda.causal_conv1d_fwd / causal_conv1d_bwd of the PyPI
// package causal-conv1d 1.1.3.post1 (not vendored in the reference; call sites
// mamba-1p1p1/mamba_ssm/modules/mamba_simple_faster.py:274-285 and
// mamba_ssm/ops/selective_scan_interface.py:496-498, 640-642, 751-753):
//
//   y[b,d,l] = act( bias[d] + sum_{k<W} w[d,k] * x[b,d,l-(W-1)+k] ),  x = 0 for negative positions.
//
// HBM-bound streaming op: lanes run along L (coalesced 64*V-element segments), one block row per
// (b, d); the W-1 halo elements come from the neighbouring lanes' segment through L1.  Backward
// recomputes the pre-activation, and reduces dw / dbias per (b, d) row in a fixed order
// (DPP wave sums, then waves in order); the sum over the batch is a second fixed-order pass
// (fv_reduce_partials) -- no float atomics.
#include "rowwalk.h"

namespace {

constexpr int CWMAX = 4;

struct ConvParams {
  const void *x, *dy;
  const float *w, *bias;     // (dim, width), (dim) or null
  void *y, *dx;
  float* part;               // (batch, dim, CWMAX + 1) per-row partials [dw (front-padded to 4) | dbias]
  int batch, dim, L, width, silu;
};

template <typename T>
__device__ __forceinline__ float ldx(const T* row, int l, int L) {
  return (l >= 0 && l < L) ? io<T>::ld(row + l) : 0.f;
}

// V consecutive outputs per lane
template <typename T, int V>
__global__ __launch_bounds__(256) void conv_bdl_fwd_kernel(ConvParams p) {
  const int d = blockIdx.y, b = blockIdx.z;
  const size_t ro = ((size_t)b * p.dim + d) * p.L;
  const T* xr = (const T*)p.x + ro;
  T* yr = (T*)p.y + ro;
  float w[CWMAX];
#pragma unroll
  for (int k = 0; k < CWMAX; ++k) {          // front-pad narrower filters with zeros: one code path for W = 2..4
    const int kk = k - (CWMAX - p.width);
    w[k] = kk >= 0 ? p.w[d * p.width + kk] : 0.f;
  }
  const float bias = p.bias ? p.bias[d] : 0.f;
  const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (l0 >= p.L) return;
  float xv[V + CWMAX - 1];
#pragma unroll
  for (int k = 0; k < V + CWMAX - 1; ++k) xv[k] = ldx(xr, l0 - (CWMAX - 1) + k, p.L);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    float a = bias;
#pragma unroll
    for (int k = 0; k < CWMAX; ++k) a = fmaf(w[k], xv[v + k], a);
    if (l0 + v < p.L) io<T>::st(yr + l0 + v, p.silu ? fv_silu(a) : a);
  }
}

// One block per (b, d) row: walks L in block-wide strips; dx[l] = sum_k w[k] * dpre[l + 3 - k].
template <typename T>
__global__ __launch_bounds__(256) void conv_bdl_bwd_kernel(ConvParams p) {
  __shared__ float s_red[4][CWMAX + 1];
  const int d = blockIdx.x, b = blockIdx.y;
  const size_t ro = ((size_t)b * p.dim + d) * p.L;
  const T* xr = (const T*)p.x + ro;
  const T* gr = (const T*)p.dy + ro;
  T* dxr = (T*)p.dx + ro;
  float w[CWMAX];
#pragma unroll
  for (int k = 0; k < CWMAX; ++k) {
    const int kk = k - (CWMAX - p.width);
    w[k] = kk >= 0 ? p.w[d * p.width + kk] : 0.f;
  }
  const float bias = p.bias ? p.bias[d] : 0.f;
  float a_w[CWMAX] = {0.f, 0.f, 0.f, 0.f}, a_b = 0.f;
  for (int l = threadIdx.x; l < p.L; l += blockDim.x) {
    // x[l-3 .. l+3], dy[l .. l+3]
    float xv[2 * CWMAX - 1], dpre[CWMAX];
#pragma unroll
    for (int k = 0; k < 2 * CWMAX - 1; ++k) xv[k] = ldx(xr, l - (CWMAX - 1) + k, p.L);
#pragma unroll
    for (int j = 0; j < CWMAX; ++j) {           // dpre[l + j]
      float a = bias;
#pragma unroll
      for (int k = 0; k < CWMAX; ++k) a = fmaf(w[k], xv[j + k], a);
      const float g = ldx(gr, l + j, p.L);
      dpre[j] = p.silu ? g * fv_silu_grad(a) : g;
    }
    float dx = 0.f;
#pragma unroll
    for (int k = 0; k < CWMAX; ++k) dx = fmaf(w[k], dpre[CWMAX - 1 - k], dx);
    io<T>::st(dxr + l, dx);
#pragma unroll
    for (int k = 0; k < CWMAX; ++k) a_w[k] = fmaf(dpre[0], xv[k], a_w[k]);   // dw[k] += dpre[l] * x[l-3+k]
    a_b += dpre[0];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CWMAX; ++k) a_w[k] = wave_sum_uniform(a_w[k]);
  a_b = wave_sum_uniform(a_b);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < CWMAX; ++k) s_red[wv][k] = a_w[k];
    s_red[wv][CWMAX] = a_b;
  }
  __syncthreads();
  if (threadIdx.x <= CWMAX) {
    float t = 0.f;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += s_red[q][threadIdx.x];
    p.part[((size_t)b * p.dim + d) * (CWMAX + 1) + threadIdx.x] = t;
  }
}

template <typename T>
int launch_conv(const ConvParams& p, int bwd, hipStream_t st) {
  if (!bwd) {
    constexpr int V = 4;
    dim3 grid(fv_cdiv(p.L, 256 * V), p.dim, p.batch);
    hipLaunchKernelGGL((conv_bdl_fwd_kernel<T, V>), grid, dim3(256), 0, st, p);
  } else {
    const int thr = p.L >= 256 ? 256 : (p.L > 64 ? 128 : 64);
    hipLaunchKernelGGL((conv_bdl_bwd_kernel<T>), dim3(p.dim, p.batch), dim3(thr), 0, st, p);
  }
  FV_LAUNCH_CHECK();
  return FV_OK;
}

int dispatch_conv(const ConvParams& p, int bwd, int dtype, hipStream_t st) {
  if (dtype == FV_F32) return launch_conv<float>(p, bwd, st);
  if (dtype == FV_BF16) return launch_conv<bf16_t>(p, bwd, st);
  return launch_conv<__half>(p, bwd, st);
}

// "Compressed scan" epilogue of the FastVim kernel fork (fastvim_kernel/mamba-1p1p1/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:68-299): out[b,d,l] = yc[b,d,l / cf] + D[d] * u[b,d,l].
template <typename T>
__global__ __launch_bounds__(256) void expand_skip_kernel(const T* __restrict__ yc, const T* __restrict__ u,
                                                          const float* __restrict__ D, T* __restrict__ out,
                                                          int dim, int L, int Lc, int cf, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const size_t bd = idx / L;
  const int l = (int)(idx - bd * L);
  const float skip = D ? D[bd % dim] * io<T>::ld(u + idx) : 0.f;
  io<T>::st(out + idx, io<T>::ld(yc + bd * Lc + l / cf) + skip);
}

int check_conv(int batch, int dim, int L, int width, int dtype) {
  FV_CHECK(batch > 0 && dim > 0 && L > 0, "causal_conv1d: empty dimension");
  FV_CHECK(width >= 2 && width <= CWMAX, "causal_conv1d only supports width between 2 and 4 (got %d)", width);
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16 || dtype == FV_F16, "causal_conv1d: unsupported dtype");
  FV_CHECK(dim <= 65535 && batch <= 65535, "causal_conv1d: batch / dim exceed the launch grid");
  return FV_OK;
}

}  // namespace

extern "C" int fv_causal_conv1d_fwd(const void* x, const float* weight, const float* bias, void* y, int batch,
                                    int dim, int seqlen, int width, int silu, int dtype, fv_stream_t stream) {
  int rc = check_conv(batch, dim, seqlen, width, dtype);
  if (rc) return rc;
  FV_CHECK(x && weight && y, "causal_conv1d_fwd: null pointer");
  ConvParams p{};
  p.x = x; p.w = weight; p.bias = bias; p.y = y;
  p.batch = batch; p.dim = dim; p.L = seqlen; p.width = width; p.silu = silu;
  return dispatch_conv(p, 0, dtype, (hipStream_t)stream);
}

extern "C" int fv_causal_conv1d_bwd(const void* x, const float* weight, const float* bias, const void* dy, void* dx,
                                    float* partials, int batch, int dim, int seqlen, int width, int silu,
                                    int dtype, fv_stream_t stream) {
  int rc = check_conv(batch, dim, seqlen, width, dtype);
  if (rc) return rc;
  FV_CHECK(x && weight && dy && dx && partials, "causal_conv1d_bwd: null pointer");
  ConvParams p{};
  p.x = x; p.w = weight; p.bias = bias; p.dy = dy; p.dx = dx; p.part = partials;
  p.batch = batch; p.dim = dim; p.L = seqlen; p.width = width; p.silu = silu;
  return dispatch_conv(p, 1, dtype, (hipStream_t)stream);
}

extern "C" int fv_scan_expand_skip_fwd(const void* yc, const void* u_full, const float* D, void* out, int batch,
                                       int dim, int seqlen, int seqlen_compressed, int dtype, fv_stream_t stream) {
  FV_CHECK(batch > 0 && dim > 0 && seqlen > 0 && seqlen_compressed > 0, "scan_expand_skip: empty dimension");
  FV_CHECK(seqlen % seqlen_compressed == 0, "scan_expand_skip: seqlen %d is not a multiple of the compressed length %d",
           seqlen, seqlen_compressed);
  FV_CHECK(yc && out && (!D || u_full), "scan_expand_skip: null pointer");
  const size_t n = (size_t)batch * dim * seqlen;
  const int cf = seqlen / seqlen_compressed;
  dim3 grid(fv_cdiv((long)n, 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FV_F32)
    hipLaunchKernelGGL(expand_skip_kernel<float>, grid, dim3(256), 0, st, (const float*)yc, (const float*)u_full, D, (float*)out, dim, seqlen, seqlen_compressed, cf, n);
  else if (dtype == FV_BF16)
    hipLaunchKernelGGL(expand_skip_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)yc, (const bf16_t*)u_full, D, (bf16_t*)out, dim, seqlen, seqlen_compressed, cf, n);
  else if (dtype == FV_F16)
    hipLaunchKernelGGL(expand_skip_kernel<__half>, grid, dim3(256), 0, st, (const __half*)yc, (const __half*)u_full, D, (__half*)out, dim, seqlen, seqlen_compressed, cf, n);
  else
    FV_CHECK(false, "scan_expand_skip: unsupported dtype");
  FV_LAUNCH_CHECK();
  return FV_OK;
}
