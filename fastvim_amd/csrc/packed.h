// Explicit 2-wide packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and packed token
// access through buffer descriptors, for the instruction-issue-bound whole-row kernels.  Files that include
// this are compiled with -fno-slp-vectorize: the pairing is written out, not left to the vectoriser.
#pragma once
#include "rowwalk.h"

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float a) { f2 o; o.x = a; o.y = a; return o; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 sigmoid2(f2 a) {
  const f2 t = a * splat(-FV_LOG2E);          // one v_pk_mul_f32 instead of two v_mul_f32
  f2 e;
  e.x = __builtin_amdgcn_exp2f(t.x);
  e.y = __builtin_amdgcn_exp2f(t.y);
  f2 d = e + 1.f, o;
  o.x = __builtin_amdgcn_rcpf(d.x);
  o.y = __builtin_amdgcn_rcpf(d.y);
  return o;
}
__device__ __forceinline__ f2 silu2(f2 a) { return a * sigmoid2(a); }

// NP channel pairs of one token, kept exactly as loaded
template <typename T, int NP> struct PairVec;
template <int NP> struct PairVec<bf16_t, NP> {     // one dword = two bf16 channels
  uint32_t w[NP];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff) { fv_buf_load_words<NP>(r, voff, soff, w); }
  __device__ __forceinline__ f2 get(int q) const {
    f2 o;
    o.x = __uint_as_float(w[q] << 16);
    o.y = __uint_as_float(w[q] & 0xffff0000u);
    return o;
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const f2 (&v)[NP]) {
    uint32_t o[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) o[q] = pack_bf16x2(v[q].x, v[q].y);
    fv_buf_store_words<NP>(r, voff, soff, o);
  }
};
template <int NP> struct PairVec<float, NP> {
  uint32_t w[2 * NP];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff) { fv_buf_load_words<2 * NP>(r, voff, soff, w); }
  __device__ __forceinline__ f2 get(int q) const {
    f2 o;
    o.x = __uint_as_float(w[2 * q]);
    o.y = __uint_as_float(w[2 * q + 1]);
    return o;
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const f2 (&v)[NP]) {
    uint32_t o[2 * NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) { o[2 * q] = __float_as_uint(v[q].x); o[2 * q + 1] = __float_as_uint(v[q].y); }
    fv_buf_store_words<2 * NP>(r, voff, soff, o);
  }
};

// (d_in, 4) fp32 conv weights of a channel pair -> one f2 per tap
__device__ __forceinline__ void load_taps2(const float* w, int c0, f2 (&t)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(w + (size_t)c0 * 4);
  const float4 b = *reinterpret_cast<const float4*>(w + (size_t)(c0 + 1) * 4);
  t[0].x = a.x; t[1].x = a.y; t[2].x = a.z; t[3].x = a.w;
  t[0].y = b.x; t[1].y = b.y; t[2].y = b.z; t[3].y = b.w;
}
__device__ __forceinline__ f2 load_f2(const float* p, int c0) { return p ? *reinterpret_cast<const f2*>(p + c0) : splat(0.f); }
