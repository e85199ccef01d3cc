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

// ---------------------------------------------------------------- buffer addressing
// A token's address is (wave-uniform token offset) + (lane channel offset).  Expressed as a flat pointer
// the compiler materialises one 64-bit VGPR pair per access; the row kernels that keep a whole row of
// loads in flight use buffer instructions instead: descriptor + uniform byte offset in SGPRs, ONE 32-bit
// lane offset VGPR shared by every access.
typedef unsigned fv_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned fv_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned fv_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fv_make_buf(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes > 0xfffff000ull ? (int)0xfffff000u : (int)bytes,
                                           0x00020000);
}

template <int W>
__device__ __forceinline__ void fv_buf_load_words(__amdgpu_buffer_rsrc_t r, int voff, int soff, uint32_t (&w)[W]) {
  if constexpr (W == 1) {
    w[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
  } else if constexpr (W == 2) {
    fv_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    w[0] = t.x; w[1] = t.y;
  } else if constexpr (W == 3) {
    fv_u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0);
    w[0] = t.x; w[1] = t.y; w[2] = t.z;
  } else if constexpr (W == 4) {
    fv_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
  } else {
    static_assert(W % 2 == 0 && W <= 8, "unsupported vector width");
    uint32_t a[W / 2], b[W / 2];
    fv_buf_load_words<W / 2>(r, voff, soff, a);
    fv_buf_load_words<W / 2>(r, voff + 2 * W, soff, b);
#pragma unroll
    for (int k = 0; k < W / 2; ++k) { w[k] = a[k]; w[W / 2 + k] = b[k]; }
  }
}

// cache policy of the row kernels' bulk stores (aux operand of the buffer-store intrinsics: bit 0 = sc0, bit 1 = nt,
// bit 4 = sc1).  0 = default.  A build-time A/B knob (FASTVIM_EXTRA_FLAGS=-DFV_BUF_STORE_AUX=<n> python -m fastvim_amd.build):
// do write-through / non-temporal stores shorten the dirty-line flush at the kernel boundary?  Measured: see DESIGN.md.
#ifndef FV_BUF_STORE_AUX
#define FV_BUF_STORE_AUX 0
#endif
template <int W>
__device__ __forceinline__ void fv_buf_store_words(__amdgpu_buffer_rsrc_t r, int voff, int soff, const uint32_t (&w)[W]) {
  if constexpr (W == 1) {
    __builtin_amdgcn_raw_buffer_store_b32(w[0], r, voff, soff, FV_BUF_STORE_AUX);
  } else if constexpr (W == 2) {
    fv_u32x2 t; t.x = w[0]; t.y = w[1];
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, FV_BUF_STORE_AUX);
  } else if constexpr (W == 3) {
    // as 8 + 4 bytes: the single buffer_store_dwordx3 form produced corrupt data here (gfx950, ROCm 7.2 hipcc)
    fv_u32x2 t; t.x = w[0]; t.y = w[1];
    __builtin_amdgcn_raw_buffer_store_b64(t, r, voff, soff, FV_BUF_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b32(w[2], r, voff + 8, soff, FV_BUF_STORE_AUX);
  } else if constexpr (W == 4) {
    fv_u32x4 t; t.x = w[0]; t.y = w[1]; t.z = w[2]; t.w = w[3];
    __builtin_amdgcn_raw_buffer_store_b128(t, r, voff, soff, FV_BUF_STORE_AUX);
  } else {
    static_assert(W % 2 == 0 && W <= 8, "unsupported vector width");
    uint32_t a[W / 2], b[W / 2];
#pragma unroll
    for (int k = 0; k < W / 2; ++k) { a[k] = w[k]; b[k] = w[W / 2 + k]; }
    fv_buf_store_words<W / 2>(r, voff, soff, a);
    fv_buf_store_words<W / 2>(r, voff + 2 * W, soff, b);
  }
}

// VEC channels of storage type T through a buffer descriptor (VEC even for bf16)
template <typename T, int VEC> struct BufIO;
template <int VEC> struct BufIO<float, VEC> {
  static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff, RawVec<float, VEC>& o) {
    uint32_t w[VEC];
    fv_buf_load_words<VEC>(r, voff, soff, w);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o.v[k] = __uint_as_float(w[k]);
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&v)[VEC]) {
    uint32_t w[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) w[k] = __float_as_uint(v[k]);
    fv_buf_store_words<VEC>(r, voff, soff, w);
  }
};
template <int VEC> struct BufIO<bf16_t, VEC> {
  static_assert(VEC == 1 || VEC % 2 == 0, "bf16 buffer access moves single channels or channel pairs");
  static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int voff, int soff, RawVec<bf16_t, VEC>& o) {
    if constexpr (VEC == 1) o.w[0] = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    else fv_buf_load_words<VEC / 2>(r, voff, soff, o.w);
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float (&v)[VEC]) {
    if constexpr (VEC == 1) {
      __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(v[0]), r, voff, soff, 0);
    } else {
      uint32_t w[VEC / 2];
#pragma unroll
      for (int k = 0; k < VEC / 2; ++k) w[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
      fv_buf_store_words<VEC / 2>(r, voff, soff, w);
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
