// hipcc-flags: -fgpu-flush-denormals-to-zero
// Kept tokens <-> pooling rows for the MAE masked mixer: a deterministic segment sum
// (replaces compute_row_means_constantdivide, mamba_simple_masked_faster.py:376-416: index_add_ over the kept
// tokens, divided by cols) and a row gather (replaces the torch.gather expansions, :281-283, 311-314).  Each is the
// other's adjoint, so the two kernels serve the forward and the backward pass.
//
// Both are tiny next to the full-length kernels (25 % of the tokens are kept): one block per output row, a thread
// owns 4 channels, the row index of a token is a scalar (wave-uniform) load, so the segment sum is a scalar-branchy
// walk over the kept tokens in token order -- fixed summation order, no atomics.
#include "common.h"
#include "rowwalk.h"

namespace {

struct RowsParams {
  const void* in;
  void* out;
  const int* idx;        // (2, B, Lk)
  int B, Lk, rows, d_in;
  size_t in_dir;         // elements between the two directions of `in`
  float scale;
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rows_segment_sum_kernel(RowsParams p) {
  const int r = blockIdx.x, b = blockIdx.y, dir = blockIdx.z;
  const int* idx = p.idx + ((size_t)dir * p.B + b) * p.Lk;
  const TI* in = (const TI*)p.in + (size_t)dir * p.in_dir + (size_t)b * p.Lk * p.d_in;
  TO* out = (TO*)p.out + (((size_t)dir * p.B + b) * p.rows + r) * p.d_in;
  for (int c = threadIdx.x * 4; c < p.d_in; c += blockDim.x * 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < p.Lk; ++t) {
      if (idx[t] == r) {          // wave-uniform
        float v[4];
        VecIO<TI, 4>::load(in + (size_t)t * p.d_in + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] *= p.scale;
    VecIO<TO, 4>::store(out + c, acc);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rows_gather_kernel(RowsParams p) {
  const int t = blockIdx.x, b = blockIdx.y, dir = blockIdx.z;
  const int r = p.idx[((size_t)dir * p.B + b) * p.Lk + t];
  const bool ok = r >= 0 && r < p.rows;
  const TI* in = (const TI*)p.in + (((size_t)dir * p.B + b) * p.rows + (ok ? r : 0)) * p.d_in;
  TO* out = (TO*)p.out + (((size_t)dir * p.B + b) * p.Lk + t) * p.d_in;
  for (int c = threadIdx.x * 4; c < p.d_in; c += blockDim.x * 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) VecIO<TI, 4>::load(in + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.scale;
    VecIO<TO, 4>::store(out + c, v);
  }
}

int check(const void* in, const int* idx, void* out, int in_dtype, int out_dtype, int B, int Lk, int rows, int d_in,
          const char* who) {
  FV_CHECK(in && idx && out, "%s: null pointer", who);
  FV_CHECK(B > 0 && Lk > 0 && rows > 0 && d_in > 0, "%s: empty dimension", who);
  FV_CHECK(d_in % 4 == 0, "%s: d_inner %d must be a multiple of 4", who, d_in);
  FV_CHECK((in_dtype == FV_F32 || in_dtype == FV_BF16) && (out_dtype == FV_F32 || out_dtype == FV_BF16),
           "%s: dtypes must be fp32 or bf16", who);
  return FV_OK;
}

#define FV_ROWS_DISPATCH(KERNEL, grid)                                                                    \
  do {                                                                                                    \
    const dim3 block(d_inner >= 1024 ? 256 : (d_inner >= 512 ? 128 : 64));                                \
    if (in_dtype == FV_F32 && out_dtype == FV_F32) hipLaunchKernelGGL((KERNEL<float, float>), grid, block, 0, st, p); \
    else if (in_dtype == FV_F32) hipLaunchKernelGGL((KERNEL<float, bf16_t>), grid, block, 0, st, p);      \
    else if (out_dtype == FV_F32) hipLaunchKernelGGL((KERNEL<bf16_t, float>), grid, block, 0, st, p);     \
    else hipLaunchKernelGGL((KERNEL<bf16_t, bf16_t>), grid, block, 0, st, p);                             \
  } while (0)

}  // namespace

extern "C" int fv_rows_segment_sum(const void* in, int in_dtype, int in_per_direction, const int* idx, void* out,
                                   int out_dtype, int batch, int n_tokens, int rows, int d_inner, float scale,
                                   fv_stream_t stream) {
  int rc = check(in, idx, out, in_dtype, out_dtype, batch, n_tokens, rows, d_inner, "rows_segment_sum");
  if (rc) return rc;
  RowsParams p{};
  p.in = in; p.out = out; p.idx = idx; p.B = batch; p.Lk = n_tokens; p.rows = rows; p.d_in = d_inner;
  p.in_dir = in_per_direction ? (size_t)batch * n_tokens * d_inner : 0;
  p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  FV_ROWS_DISPATCH(rows_segment_sum_kernel, dim3(rows, batch, 2));
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_rows_gather(const void* in, int in_dtype, const int* idx, void* out, int out_dtype, int batch,
                              int n_tokens, int rows, int d_inner, float scale, fv_stream_t stream) {
  int rc = check(in, idx, out, in_dtype, out_dtype, batch, n_tokens, rows, d_inner, "rows_gather");
  if (rc) return rc;
  RowsParams p{};
  p.in = in; p.out = out; p.idx = idx; p.B = batch; p.Lk = n_tokens; p.rows = rows; p.d_in = d_inner;
  p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  FV_ROWS_DISPATCH(rows_gather_kernel, dim3(n_tokens, batch, 2));
  FV_LAUNCH_CHECK();
  return FV_OK;
}
