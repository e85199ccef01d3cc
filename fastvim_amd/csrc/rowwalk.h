// Channel-last "row walker" helpers shared by the fused mixer kernels.
//
// Activations are token-major / channel-last: xz is (B, L, 2*d_in) with the x half and the z
// half of a token adjacent, everything else (B, L, d_in).  A wave owns one chunk of
// 64*VEC channels of one pooling row (b, i) and walks the `cols` tokens of that row in
// sequence order; every global access of a wave is one contiguous 64*VEC*sizeof(T) segment.
//
// The token grid is never transposed in memory.  The odd-layer "rotate" of the reference
// (models/fastvim.py:192-210) is a pair of token strides: sequence position (i, j) of the
// mixer lives at memory token  i*s_i + j*s_j  (even layers: s_i = cols, s_j = 1; odd layers,
// whose mixer sees the transposed grid: s_i = 1, s_j = rows).
#pragma once
#include "common.h"

struct Geo {
  int rows, cols, L;   // pooling rows; FLAT positions per row (= patch columns * tokens_per_patch); L = rows*cols
  int s_i, s_j;        // memory strides, in patches, of (row, patch column) of the sequence grid
  int tpp, pcols;      // tokens per patch (channel-wise tokenization, Channel-First order), patch columns
};
// Sequence position s = (i*pcols + j)*tpp + c (row i, patch column j, channel token c); the pooling
// group of a token is (i, c) = pooled index i*tpp + c  (mamba_simple_channel_faster.py:242-256).

static inline Geo make_geo(int rows, int pcols, int s_i, int s_j, int tpp) {
  return Geo{rows, pcols * tpp, rows * pcols * tpp, s_i, s_j, tpp, pcols};
}

// TP = false: compiled for tokens_per_patch == 1 (the FastVim models) without the channel-slot decode
template <bool TP = true>
__device__ __forceinline__ int tok_mem(const Geo& g, int s) {
  if (g.s_j == 1) return s;          // natural order: sequence position == memory token
  const int i = s / g.cols;
  const int jf = s - i * g.cols;
  if constexpr (!TP) {
    return i * g.s_i + jf * g.s_j;
  } else {
    const int j = jf / g.tpp;
    const int c = jf - j * g.tpp;
    return (i * g.s_i + j * g.s_j) * g.tpp + c;
  }
}

// ---------------------------------------------------------------- vector I/O of VEC channels
template <int N> struct raw_words;   // N 32-bit words with 4-byte alignment
template <> struct raw_words<1> { typedef uint32_t type __attribute__((aligned(4))); };
template <> struct raw_words<2> { typedef uint32_t type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct raw_words<3> { typedef uint32_t type __attribute__((ext_vector_type(3), aligned(4))); };
template <> struct raw_words<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };

template <typename T, int VEC> struct VecIO;

template <int VEC> struct VecIO<float, VEC> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
      v[0] = *p;
    } else if constexpr (VEC % 4 == 0) {
#pragma unroll
      for (int k = 0; k < VEC / 4; ++k) {
        auto w = *reinterpret_cast<const typename raw_words<4>::type*>(p + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * k + e] = __uint_as_float(w[e]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC / 2; ++k) {
        auto w = *reinterpret_cast<const typename raw_words<2>::type*>(p + 2 * k);
        v[2 * k] = __uint_as_float(w[0]);
        v[2 * k + 1] = __uint_as_float(w[1]);
      }
    }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
      *p = v[0];
    } else if constexpr (VEC % 4 == 0) {
#pragma unroll
      for (int k = 0; k < VEC / 4; ++k) {
        typename raw_words<4>::type w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(v[4 * k + e]);
        *reinterpret_cast<typename raw_words<4>::type*>(p + 4 * k) = w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < VEC / 2; ++k) {
        typename raw_words<2>::type w;
        w[0] = __float_as_uint(v[2 * k]);
        w[1] = __float_as_uint(v[2 * k + 1]);
        *reinterpret_cast<typename raw_words<2>::type*>(p + 2 * k) = w;
      }
    }
  }
};

template <int VEC> struct VecIO<bf16_t, VEC> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[VEC]) {
    if constexpr (VEC == 1) {
      v[0] = io<bf16_t>::ld(p);
    } else {
      constexpr int W = VEC / 2;
      auto w = *reinterpret_cast<const typename raw_words<W>::type*>(p);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        uint32_t x;
        if constexpr (W == 1) x = w; else x = w[k];
        v[2 * k] = __uint_as_float(x << 16);
        v[2 * k + 1] = __uint_as_float(x & 0xffff0000u);
      }
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
      io<bf16_t>::st(p, v[0]);
    } else {
      constexpr int W = VEC / 2;
      typename raw_words<W>::type w;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        uint32_t x = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        if constexpr (W == 1) w = x; else w[k] = x;
      }
      *reinterpret_cast<typename raw_words<W>::type*>(p) = w;
    }
  }
};

// VEC channels held exactly as loaded (bf16 stays packed: half the registers while a load is in flight)
template <typename T, int VEC> struct RawVec;
template <int VEC> struct RawVec<float, VEC> {
  float v[VEC];
  __device__ __forceinline__ void load(const float* p) { VecIO<float, VEC>::load(p, v); }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
  }
  __device__ __forceinline__ void get(float (&o)[VEC]) const {
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = v[k];
  }
};
template <int VEC> struct RawVec<bf16_t, VEC> {
  static constexpr int W = (VEC + 1) / 2;
  uint32_t w[W];
  __device__ __forceinline__ void load(const bf16_t* p) {
    if constexpr (VEC == 1) {
      w[0] = *reinterpret_cast<const uint16_t*>(p);
    } else {
      auto r = *reinterpret_cast<const typename raw_words<W>::type*>(p);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if constexpr (W == 1) w[k] = r; else w[k] = r[k];
      }
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = 0u;
  }
  __device__ __forceinline__ void get(float (&o)[VEC]) const {
    if constexpr (VEC == 1) {
      o[0] = __uint_as_float(w[0] << 16);
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        o[2 * k] = __uint_as_float(w[k] << 16);
        o[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
      }
    }
  }
};

template <int VEC>
__device__ __forceinline__ void load_f32(const float* p, float (&v)[VEC]) {
  VecIO<float, VEC>::load(p, v);
}

// ---------------------------------------------------------------- wave reduction via DPP
// Sum over the 64 lanes; result returned to every lane (through an SGPR).
__device__ __forceinline__ float wave_sum_uniform(float v) {
  int x = __float_as_int(v);
#define FV_DPP_ADD(ctrl, rmask)                                                              \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true))
  FV_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  FV_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  FV_DPP_ADD(0x141, 0xf);  // row_half_mirror
  FV_DPP_ADD(0x140, 0xf);  // row_mirror  -> every lane holds its 16-lane row sum
  FV_DPP_ADD(0x142, 0xa);  // row_bcast15 -> rows 1,3 add the previous row's sum
  FV_DPP_ADD(0x143, 0xc);  // row_bcast31 -> rows 2,3 add lane 31's running sum
#undef FV_DPP_ADD
  (void)x;
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
