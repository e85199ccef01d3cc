// Fused AdamW over the flat parameter buffer (+ optional EMA of the weights, + bf16 shadow refresh).
// Replaces, per training step of the reference (imagenet_classification/supervised_imagenet.py:134-147,
// 270-276): torch.optim.AdamW over two parameter groups, ModelEmaV2.update (a full-parameter lerp) and
// the per-layer autocast weight casts -- ~20 multi-tensor launches + 6 casts per block -- by ONE
// HBM-bound pass: 16 B read + 14 B written per parameter (+ 8 B with EMA).
// `lr` and `step` live in device memory so a captured HIP graph replays with a changing schedule.
#include "common.h"

namespace {

struct AdamParams {
  float *p, *m, *v, *ema;
  const float* g;
  bf16_t* shadow;
  const uint8_t* decay_mask;   // 1 = apply weight decay
  const float* lr;             // device scalar
  float* step;                 // device scalar (float), incremented by this launch
  float beta1, beta2, eps, weight_decay, ema_decay;
  float grad_scale;            // every gradient element is multiplied by this as it is read (1 / world size after a sum all-reduce)
  size_t n;
};

__global__ __launch_bounds__(256) void adamw_flat_kernel(AdamParams a) {
  const float t = a.step[0] + 1.f;                     // every thread reads the pre-increment value
  const float lr = a.lr[0];
  const float bc1 = 1.f - __powf(a.beta1, t), bc2 = 1.f - __powf(a.beta2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < a.n; i += stride) {
    float4 p = *reinterpret_cast<const float4*>(a.p + i);
    float4 g = *reinterpret_cast<const float4*>(a.g + i);
    g.x *= a.grad_scale; g.y *= a.grad_scale; g.z *= a.grad_scale; g.w *= a.grad_scale;
    float4 m = *reinterpret_cast<const float4*>(a.m + i);
    float4 v = *reinterpret_cast<const float4*>(a.v + i);
    const uint32_t mask = *reinterpret_cast<const uint32_t*>(a.decay_mask + i);
    float* pp = &p.x; const float* gg = &g.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((mask >> (8 * e)) & 1u) pp[e] *= 1.f - lr * a.weight_decay;      // decoupled weight decay
      mm[e] = a.beta1 * mm[e] + (1.f - a.beta1) * gg[e];
      vv[e] = a.beta2 * vv[e] + (1.f - a.beta2) * gg[e] * gg[e];
      pp[e] -= step_size * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + a.eps);
    }
    *reinterpret_cast<float4*>(a.p + i) = p;
    *reinterpret_cast<float4*>(a.m + i) = m;
    *reinterpret_cast<float4*>(a.v + i) = v;
    if (a.shadow) {
      uint2 pk = {pack_bf16x2(p.x, p.y), pack_bf16x2(p.z, p.w)};
      *reinterpret_cast<uint2*>(a.shadow + i) = pk;
    }
    if (a.ema) {
      float4 e4 = *reinterpret_cast<const float4*>(a.ema + i);
      e4.x = a.ema_decay * e4.x + (1.f - a.ema_decay) * p.x;
      e4.y = a.ema_decay * e4.y + (1.f - a.ema_decay) * p.y;
      e4.z = a.ema_decay * e4.z + (1.f - a.ema_decay) * p.z;
      e4.w = a.ema_decay * e4.w + (1.f - a.ema_decay) * p.w;
      *reinterpret_cast<float4*>(a.ema + i) = e4;
    }
  }
}

__global__ void bump_step_kernel(float* step) { step[0] += 1.f; }

}  // namespace

extern "C" int fv_adamw_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema,
                             void* shadow_bf16, const uint8_t* decay_mask, const float* lr, float* step,
                             float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                             size_t n, fv_stream_t stream) {
  FV_CHECK(params && grads && exp_avg && exp_avg_sq && decay_mask && lr && step, "adamw_flat: null pointer");
  FV_CHECK(n % 4 == 0, "adamw_flat: element count must be a multiple of 4 (pad the flat buffer)");
  AdamParams a{};
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.shadow = (bf16_t*)shadow_bf16;
  a.decay_mask = decay_mask; a.lr = lr; a.step = step;
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.ema_decay = ema_decay; a.grad_scale = grad_scale; a.n = n;
  if (n == 0) return FV_OK;
  long blocks = fv_cdiv((long)(n / 4), 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((int)blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, st, step);   // after: the main pass read the old value
  FV_LAUNCH_CHECK();
  return FV_OK;
}
