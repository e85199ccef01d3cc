// Shared device helpers for the FastVim gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>

#include "../../include/fastvim_hip.h"

#define FV_WAVE 64
#define FV_LOG2E 1.4426950408889634f

typedef __hip_bfloat16 bf16_t;

// ---------------------------------------------------------------- errors
void fv_set_error(const char* fmt, ...);
#define FV_CHECK(cond, ...)                  \
  do {                                       \
    if (!(cond)) {                           \
      fv_set_error(__VA_ARGS__);             \
      return FV_ERR_INVALID;                 \
    }                                        \
  } while (0)
#define FV_LAUNCH_CHECK()                                                 \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      fv_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                   hipGetErrorString(e_));                                \
      return FV_ERR_HIP;                                                  \
    }                                                                     \
  } while (0)

static inline int fv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: a launcher's "already done" flag is kept per
// device (one bit each), so a process that drives several GPUs sets it on every one of them.  A race between two host
// threads only repeats an idempotent call.
struct FvOncePerDevice {
  std::atomic<unsigned long long> seen{0};
  bool first() {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    return (seen.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;     // exactly one caller per device sees "first"
  }
};

// CU count of the CURRENT device (kept per device: a process may drive several GPUs, and they need not be the same part)
static inline int fv_cu_count() {
  static std::atomic<int> cus[64];
  int d = 0;
  (void)hipGetDevice(&d);
  int c = cus[d & 63].load(std::memory_order_relaxed);
  if (c <= 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
    cus[d & 63].store(c, std::memory_order_relaxed);
  }
  return c;
}

// A/B hooks of the kernel dispatchers.  The shipped library takes the measured default of every choice and reads NO
// environment variable; a build with -DFASTVIM_TUNING_HOOKS (python -m fastvim_amd.build --tuning) reads FASTVIM_<NAME>
// once per process so that an experiment can be repeated on one box in one call (tools/README.md lists them).
#ifdef FASTVIM_TUNING_HOOKS
#include <stdlib.h>
static inline int fv_tune(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static inline int fv_tune(const char*, int dflt) { return dflt; }
#endif

// ---------------------------------------------------------------- scalar type traits
template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) {
    return __uint_as_float(((uint32_t) * reinterpret_cast<const uint16_t*>(p)) << 16);
  }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = __float2bfloat16(v); }
};
template <> struct io<__half> {
  static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }
};

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even via the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float v) {
  bf16_t h = __float2bfloat16(v);
  return *reinterpret_cast<uint16_t*>(&h);
}
// two floats -> one dword of bf16 (lo in the low half): ONE v_cvt_pk_bf16_f32 (the scalar form above, twice, plus the
// merge is three instructions)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef float fv_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 fv_bf16x2 __attribute__((ext_vector_type(2)));
  const fv_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fv_bf16x2));
}

// ---------------------------------------------------------------- math
__device__ __forceinline__ float fv_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fv_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fv_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float fv_silu(float x) { return x / (1.0f + __expf(-x)); }
// d/dx silu(x) = s * (1 + x * (1 - s))
__device__ __forceinline__ float fv_silu_grad(float x) {
  float s = fv_sigmoid(x);
  return s * (1.0f + x * (1.0f - s));
}
// log(1 + e) for e in [0, 1]: short series below 1/16 (rel. error < 1e-8), hardware log above
__device__ __forceinline__ float fv_log1p_unit(float e) {
  const float ser = e * (1.f - e * (0.5f - e * (0.33333334f - e * (0.25f - e * (0.2f - e * 0.16666667f)))));
  return e < 0.0625f ? ser : __logf(1.f + e);
}
// softplus(x) = max(x, 0) + log1p(exp(-|x|)); branch-free, accurate to ~1e-7 relative for the small
// step sizes (1e-3 .. 1e-1) the dt parameterisation produces.  The reference's threshold form
// (x > 20 -> x, selective_scan_fwd_kernel.cuh:153-156) differs from this by < 2.1e-9.
__device__ __forceinline__ float fv_softplus(float x) {
  return fmaxf(x, 0.f) + fv_log1p_unit(__expf(-fabsf(x)));
}

// ---------------------------------------------------------------- wave64 collectives
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
