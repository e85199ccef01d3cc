// Soft-target cross-entropy, forward and gradient in one launch.
// Replaces, per training step of the reference (imagenet_classification/supervised_imagenet.py:83, 109-115:
// timm.loss.SoftTargetCrossEntropy under mixup / label smoothing): logits.float(), log_softmax, mul, neg, sum, mean
// and their five autograd kernels --
//     loss_b = sum_c -t[b][c] * log_softmax(x[b])[c],      loss = mean_b loss_b
//     d loss / d x[b][c] = (softmax(x[b])[c] * sum_c' t[b][c'] - t[b][c]) / B
// One wave per row: the row lives in registers (C <= 64 * 32), max and sums are DPP wave reductions, the gradient is
// written in the same pass.  Rows are summed to the scalar loss by a single-wave second kernel in a fixed order.
#include "common.h"

namespace {

constexpr int EPL = 32;      // elements per lane: rows up to 2048 classes

template <typename T>
__global__ __launch_bounds__(256) void soft_ce_rows_kernel(const T* __restrict__ x, const float* __restrict__ t,
                                                            float* __restrict__ loss_rows, float* __restrict__ dx, int B,
                                                            int C, float inv_b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const T* xr = x + (size_t)row * C;
  const float* tr = t + (size_t)row * C;
  float xv[EPL], tv[EPL];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int c = k * 64 + lane;
    xv[k] = c < C ? io<T>::ld(xr + c) : -3.0e38f;
    tv[k] = c < C ? tr[c] : 0.f;
    mx = fmaxf(mx, xv[k]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int c = k * 64 + lane;
    const float e = c < C ? __expf(xv[k] - mx) : 0.f;
    se += e;
    st += tv[k];
    stx += c < C ? tv[k] * (xv[k] - mx) : 0.f;
    xv[k] = e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o);
    st += __shfl_xor(st, o);
    stx += __shfl_xor(stx, o);
  }
  // sum_c -t (x - mx - log se) = st * log se - sum_c t (x - mx)
  if (lane == 0) loss_rows[row] = st * __logf(se) - stx;
  const float rs = st / se;
  float* dr = dx + (size_t)row * C;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int c = k * 64 + lane;
    if (c < C) dr[c] = (xv[k] * rs - tv[k]) * inv_b;
  }
}

// loss = inv_b * sum_b loss_rows[b], one wave, fixed order (lane-strided partial sums, then the butterfly)
__global__ __launch_bounds__(64) void soft_ce_mean_kernel(const float* __restrict__ loss_rows, float* __restrict__ loss,
                                                          int B, float inv_b) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += loss_rows[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) loss[0] = s * inv_b;
}

}  // namespace

extern "C" int fv_soft_target_ce(const void* logits, int logits_dtype, const float* target, float* loss_rows, float* loss,
                                 float* dlogits, int batch, int classes, fv_stream_t stream) {
  FV_CHECK(logits && target && loss_rows && loss && dlogits, "soft_target_ce: null pointer");
  FV_CHECK(batch > 0 && classes > 0, "soft_target_ce: empty dimension");
  FV_CHECK(classes <= 64 * EPL, "soft_target_ce: at most %d classes (got %d)", 64 * EPL, classes);
  FV_CHECK(logits_dtype == FV_F32 || logits_dtype == FV_BF16, "soft_target_ce: logits must be fp32 or bf16");
  hipStream_t st = (hipStream_t)stream;
  const float inv_b = 1.f / (float)batch;
  const dim3 grid(fv_cdiv(batch, 4)), block(256);
  if (logits_dtype == FV_F32)
    hipLaunchKernelGGL(soft_ce_rows_kernel<float>, grid, block, 0, st, (const float*)logits, target, loss_rows, dlogits, batch, classes, inv_b);
  else
    hipLaunchKernelGGL(soft_ce_rows_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, target, loss_rows, dlogits, batch, classes, inv_b);
  hipLaunchKernelGGL(soft_ce_mean_kernel, dim3(1), dim3(64), 0, st, loss_rows, loss, batch, inv_b);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
