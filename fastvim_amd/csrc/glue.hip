// Small memory-bound ops around the backbone that the training step would otherwise run as strings of library
// elementwise / reduce launches (each a 2.5-5 us node of the captured step):
//   * patch unfold + cast: the k == stride patch-embed Conv2d (models/fastvim.py:95) as a GEMM operand;
//   * token mean pool and its adjoint (models/fastvim.py:529-531, final_pool_type == "mean");
//   * the stochastic-depth keep table (timm DropPath: floor(keep + U) / keep, one row per DropPath module);
//   * scale-by-a-device-scalar + cast (the loss gradient handed to the head), column sums (head bias gradient).
#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Vec4;      // four consecutive elements <-> four floats
template <> struct Vec4<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[4]) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) {
    u32x2 t;
    t.x = pack_bf16x2(v[0], v[1]);
    t.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = t;
  }
};

// ---- patch unfold: out[b][gi*gw + gj][(c*ph + pi)*pw + pj] = img[b][c][gi*ph + pi][gj*pw + pj] ------------------
// One workgroup per (chunk of GJ patches, patch row, image): image rows come in as 16-byte segments (coalesced along
// W), are converted and placed at their position inside the patch row in LDS; the GJ patch rows are one contiguous
// stretch of `out` and leave as 16-byte stores.
constexpr int UNF_GJ = 16;

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patch_unfold_kernel(const TI* __restrict__ img, TO* __restrict__ out, int C, int H,
                                                           int W, int ph, int pw, int gw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TO* tile = reinterpret_cast<TO*>(smem);
  const int gj0 = blockIdx.x * UNF_GJ, gi = blockIdx.y, b = blockIdx.z;
  const int gjn = min(UNF_GJ, gw - gj0);
  const int Kp = C * ph * pw;                 // elements per patch row
  const int xv = gjn * pw / 4;                // 4-element vectors per image-row segment of the chunk
  const TI* src = img + ((size_t)b * C * H + (size_t)gi * ph) * W + (size_t)gj0 * pw;
  for (int e = threadIdx.x; e < C * ph * xv; e += blockDim.x) {
    const int row = e / xv, x = (e - row * xv) * 4;      // row = c * ph + pi
    const int c = row / ph, pi = row - c * ph;
    float v[4];
    Vec4<TI>::ld(src + ((size_t)c * H + pi) * W + x, v);
    const int gj = x / pw, pj = x - gj * pw;
    Vec4<TO>::st(tile + (size_t)gj * Kp + row * pw + pj, v);
  }
  __syncthreads();
  TO* dst = out + (((size_t)b * gridDim.y + gi) * gw + gj0) * Kp;
  constexpr int EV = 16 / sizeof(TO);
  const int nv = gjn * Kp / EV;
  for (int e = threadIdx.x; e < nv; e += blockDim.x)
    reinterpret_cast<u32x4*>(dst)[e] = reinterpret_cast<const u32x4*>(tile)[e];
}

// ---- token mean pool: out[b][d] = (1/L) sum_l x[b][l][d] ------------------------------------------------------------
// One 16-wave workgroup per (batch element, 4*64-channel slab): lane = 4 channels, the waves split the tokens (four
// loads in flight each -- with 4 waves and one load in flight the 196-token sum was 49 dependent round trips, 15 us),
// fixed-order sum through LDS.
constexpr int MP_WAVES = 16;
template <typename T>
__global__ __launch_bounds__(64 * MP_WAVES) void mean_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int L, int D, float inv) {
  __shared__ float s_acc[MP_WAVES][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int d = (blockIdx.x * 64 + lane) * 4, b = blockIdx.y;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (d < D) {
    const T* xp = x + (size_t)b * L * D + d;
    int l = wv;
    for (; l + 3 * MP_WAVES < L; l += 4 * MP_WAVES) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) Vec4<T>::ld(xp + (size_t)(l + u * MP_WAVES) * D, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += v[u][k];
    }
    for (; l < L; l += MP_WAVES) {
      float v[4];
      Vec4<T>::ld(xp + (size_t)l * D, v);
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) s_acc[wv][lane * 4 + k] = a[k];
  __syncthreads();
  if (wv == 0 && d < D) {
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < MP_WAVES; ++w) t += s_acc[w][lane * 4 + k];
      r[k] = t * inv;
    }
    Vec4<T>::st(out + (size_t)b * D + d, r);
  }
}

// dx[b][l][d] = g[b][d] * (1/L): the product is formed in fp32 and rounded once, as the library's expand / div does
// (inv = 1/L comes from the host: under -ffast-math a device-side 1.f / L is the approximate reciprocal)
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, int L, int D,
                                                            int rows_per_block, float inv) {
  const int b = blockIdx.y, dv = D / 4;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  for (int e = threadIdx.x; e < (l1 - l0) * dv; e += blockDim.x) {
    const int l = l0 + e / dv, d = (e % dv) * 4;
    float v[4];
    Vec4<T>::ld(g + (size_t)b * D + d, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= inv;
    Vec4<T>::st(dx + ((size_t)b * L + l) * D + d, v);
  }
}

__global__ __launch_bounds__(256) void droppath_table_kernel(float* __restrict__ table, const float* __restrict__ keep,
                                                             const float* __restrict__ inv, int mods, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mods * batch) {
    const int m = i / batch;
    table[i] = floorf(table[i] + keep[m]) * inv[m];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         T* __restrict__ y, size_t n) {
  const float s = scale[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    io<T>::st(y + i, x[i] * s);
}

// out[c] (+)= sum_r x[r][c]: lane = column, the 4 waves of a workgroup split the rows, fixed-order sum through LDS
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                         int accumulate) {
  __shared__ float s_acc[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
  float a = 0.f;
  if (c < cols)
    for (int r = wv; r < rows; r += 4) a += io<T>::ld(x + (size_t)r * cols + c);
  s_acc[wv][lane] = a;
  __syncthreads();
  if (wv == 0 && c < cols) {
    const float t = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

bool dt_ok(int dt) { return dt == FV_F32 || dt == FV_BF16; }


// Transposed bf16 shadows of up to 64 equal-shape weights in one launch: dst[j] (cols, rows) = src[j] (rows, cols)^T,
// 32 x 32 tiles through LDS (both sides coalesced).  in_proj.weight (2 d_inner, d_model) -> (d_model, 2 d_inner): the
// K-contiguous operand fv_mixer_conv_pool_bwd_dgrad streams into MFMA registers.
constexpr int TRJ_MAX = 64;
struct TransposeJobs {
  const uint16_t* src[TRJ_MAX];
  uint16_t* dst[TRJ_MAX];
  int rows, cols;
};
__global__ __launch_bounds__(256) void transpose_bf16_kernel(TransposeJobs J) {
  __shared__ uint16_t t[32][33];
  const uint16_t* src = J.src[blockIdx.z];
  uint16_t* dst = J.dst[blockIdx.z];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    t[ty + 8 * i][tx] = (r < J.rows && c < J.cols) ? src[(size_t)r * J.cols + c] : (uint16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < J.rows && c < J.cols) dst[(size_t)c * J.rows + r] = t[tx][ty + 8 * i];
  }
}

}  // namespace

extern "C" int fv_patch_unfold(const void* img, int img_dtype, void* out, int out_dtype, int batch, int chans, int height,
                               int width, int ph, int pw, fv_stream_t stream) {
  FV_CHECK(img && out, "patch_unfold: null pointer");
  FV_CHECK(dt_ok(img_dtype) && dt_ok(out_dtype), "patch_unfold: dtypes must be fp32 or bf16");
  FV_CHECK(batch > 0 && chans > 0 && ph > 0 && pw > 0 && height >= ph && width >= pw, "patch_unfold: empty dimension");
  FV_CHECK(height % ph == 0 && width % pw == 0, "patch_unfold: image %dx%d is not whole %dx%d patches", height, width, ph, pw);
  FV_CHECK(pw % 8 == 0 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 15) == 0,
           "patch_unfold: patch width must be a multiple of 8 and the buffers 16-byte aligned");
  const int gh = height / ph, gw = width / pw;
  const size_t osz = out_dtype == FV_F32 ? 4 : 2;
  const size_t smem = (size_t)UNF_GJ * chans * ph * pw * osz;
  FV_CHECK(smem <= 64 * 1024, "patch_unfold: %d x %d x %d patches do not fit the staging tile", chans, ph, pw);
  FV_CHECK(gh <= 65535 && batch <= 65535, "patch_unfold: grid too large");
  const dim3 grid(fv_cdiv(gw, UNF_GJ), gh, batch), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FV_UNF(TI, TO) hipLaunchKernelGGL((patch_unfold_kernel<TI, TO>), grid, block, smem, st, (const TI*)img, (TO*)out, chans, height, width, ph, pw, gw)
  if (img_dtype == FV_F32 && out_dtype == FV_BF16) FV_UNF(float, bf16_t);
  else if (img_dtype == FV_F32) FV_UNF(float, float);
  else if (out_dtype == FV_BF16) FV_UNF(bf16_t, bf16_t);
  else FV_UNF(bf16_t, float);
#undef FV_UNF
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_mean_pool_fwd(const void* x, void* out, int batch, int tokens, int dim, int dtype, fv_stream_t stream) {
  FV_CHECK(x && out, "mean_pool_fwd: null pointer");
  FV_CHECK(dt_ok(dtype), "mean_pool_fwd: dtype must be fp32 or bf16");
  FV_CHECK(batch > 0 && tokens > 0 && dim > 0 && dim % 4 == 0 && batch <= 65535, "mean_pool_fwd: bad shape (%d, %d, %d)", batch, tokens, dim);
  const dim3 grid(fv_cdiv(dim, 256), batch), block(64 * MP_WAVES);
  hipStream_t st = (hipStream_t)stream;
  const float inv = 1.f / (float)tokens;
  if (dtype == FV_F32) hipLaunchKernelGGL(mean_pool_fwd_kernel<float>, grid, block, 0, st, (const float*)x, (float*)out, tokens, dim, inv);
  else hipLaunchKernelGGL(mean_pool_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, (bf16_t*)out, tokens, dim, inv);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_mean_pool_bwd(const void* g, void* dx, int batch, int tokens, int dim, int dtype, fv_stream_t stream) {
  FV_CHECK(g && dx, "mean_pool_bwd: null pointer");
  FV_CHECK(dt_ok(dtype), "mean_pool_bwd: dtype must be fp32 or bf16");
  FV_CHECK(batch > 0 && tokens > 0 && dim > 0 && dim % 4 == 0 && batch <= 65535, "mean_pool_bwd: bad shape (%d, %d, %d)", batch, tokens, dim);
  const int rpb = fv_cdiv(tokens, 8);
  const dim3 grid(fv_cdiv(tokens, rpb), batch), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float inv = 1.f / (float)tokens;
  if (dtype == FV_F32) hipLaunchKernelGGL(mean_pool_bwd_kernel<float>, grid, block, 0, st, (const float*)g, (float*)dx, tokens, dim, rpb, inv);
  else hipLaunchKernelGGL(mean_pool_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)g, (bf16_t*)dx, tokens, dim, rpb, inv);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_droppath_table(float* table, const float* keep, const float* inv_keep, int mods, int batch, fv_stream_t stream) {
  FV_CHECK(table && keep && inv_keep && mods > 0 && batch > 0, "droppath_table: bad arguments");
  hipLaunchKernelGGL(droppath_table_kernel, dim3(fv_cdiv((long)mods * batch, 256)), dim3(256), 0, (hipStream_t)stream, table, keep,
                     inv_keep, mods, batch);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_scale_cast(const float* x, const float* scale, void* y, int y_dtype, size_t n, fv_stream_t stream) {
  FV_CHECK(x && scale && y && n > 0, "scale_cast: bad arguments");
  FV_CHECK(dt_ok(y_dtype), "scale_cast: output must be fp32 or bf16");
  const dim3 grid(fv_cdiv((long)n, 256) < 2048 ? fv_cdiv((long)n, 256) : 2048), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (y_dtype == FV_F32) hipLaunchKernelGGL(scale_cast_kernel<float>, grid, block, 0, st, x, scale, (float*)y, n);
  else hipLaunchKernelGGL(scale_cast_kernel<bf16_t>, grid, block, 0, st, x, scale, (bf16_t*)y, n);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_column_sum(const void* x, int dtype, float* out, int rows, int cols, int accumulate, fv_stream_t stream) {
  FV_CHECK(x && out && rows > 0 && cols > 0, "column_sum: bad arguments");
  FV_CHECK(dt_ok(dtype), "column_sum: input must be fp32 or bf16");
  const dim3 grid(fv_cdiv(cols, 64)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FV_F32) hipLaunchKernelGGL(column_sum_kernel<float>, grid, block, 0, st, (const float*)x, out, rows, cols, accumulate);
  else hipLaunchKernelGGL(column_sum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, out, rows, cols, accumulate);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_transpose_bf16_batched(const void* const* srcs, void* const* dsts, int njobs, int rows, int cols,
                                         fv_stream_t stream) {
  FV_CHECK(srcs && dsts && njobs > 0 && njobs <= TRJ_MAX, "transpose_bf16_batched: 1..%d jobs", TRJ_MAX);
  FV_CHECK(rows > 0 && cols > 0 && fv_cdiv(rows, 32) <= 65535, "transpose_bf16_batched: bad shape (%d, %d)", rows, cols);
  TransposeJobs J{};
  for (int j = 0; j < njobs; ++j) {
    FV_CHECK(srcs[j] && dsts[j], "transpose_bf16_batched: null pointer in job %d", j);
    J.src[j] = (const uint16_t*)srcs[j];
    J.dst[j] = (uint16_t*)dsts[j];
  }
  J.rows = rows; J.cols = cols;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(fv_cdiv(cols, 32), fv_cdiv(rows, 32), njobs), dim3(256), 0, (hipStream_t)stream, J);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
