// Tile staging / fragment helpers of the bf16 MFMA GEMMs, shared by csrc/gemm_mfma.hip and the fused producer kernels
// (csrc/convpool_dgrad.hip): LDS-DMA plans, fragment reads for K-contiguous and K-slow operands, the second GEMM phase of
// a fused launch.  Everything lives in an anonymous namespace: one copy per translation unit.
#pragma once
#include <stdlib.h>
#include <utility>

#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
enum { KC = 0, KS = 1 };

// KS tiles: rows (k) are EXT*2 bytes = a multiple of the 256-byte bank row, and a transposing read
// touches 8 different k rows at one column per 32-lane half -> 8-way conflict.  XOR the 32-byte
// column chunk with a per-row code so those 8 rows land on 8 different chunks.
template <int EXT>
__device__ __forceinline__ int ks_swz(int k) {
  const int code = (k & 3) | (((k >> 3) & 1) << 2);
  // 192-wide tiles have 12 32-byte chunks = 3 groups of 4: XOR inside a group only
  // 96-wide tiles have 6 chunks = 3 groups of 2
  return EXT == 192 ? (code & 3) : EXT == 96 ? (code & 1) : (EXT >= 128 ? code : (code & (EXT / 16 - 1)));
}

// ---- staging: 256 threads move a (ROWS x 64) KC tile or a (64 x COLS) KS tile, 16 B per access ----
template <int MODE, int EXT, int NT = 256>   // EXT = tile extent along the non-K dim (64 .. 256), NT = threads per block
struct Stage {
  static constexpr int NV = EXT * BK / 8 / NT;   // 16-byte vectors per thread
  u32x4 v[NV];
  // global -> registers.  r0 = first row/col of the tile in the non-K dim, k0 = first k.
  __device__ __forceinline__ void load(const bf16_t* base, long ld, int r0, int k0, int rmax, int kmax, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NT;
      int r, k;
      if (MODE == KC) { r = e >> 3; k = (e & 7) * 8; }                 // 8 vectors per 64-k row
      else { k = e / (EXT / 8); r = (e % (EXT / 8)) * 8; }             // EXT/8 vectors per k row
      const int gr = r0 + r, gk = k0 + k;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (MODE == KC) {
        v[i] = (gr < rmax && gk < kmax) ? *reinterpret_cast<const u32x4*>(base + (long)gr * ld + gk) : z;
      } else {
        v[i] = (gk < kmax && gr < rmax) ? *reinterpret_cast<const u32x4*>(base + (long)gk * ld + gr) : z;
      }
    }
  }
  // registers -> LDS.  KC: [row][64] with chunk ^= row & 7;  KS: [k][EXT] as is.
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NT;
      int off;
      if (MODE == KC) {
        const int r = e >> 3, c = e & 7;
        off = r * 128 + ((c ^ (r & 7)) << 4);
      } else {
        const int k = e / (EXT / 8), cb = (e % (EXT / 8)) * 16;          // byte column of this 16-B vector
        off = k * (EXT * 2) + ((((cb >> 5) ^ ks_swz<EXT>(k)) << 5) | (cb & 16));
      }
      *reinterpret_cast<u32x4*>(lds + off) = v[i];
    }
  }
};

// ---- LDS-DMA staging (global_load_lds_dwordx4): the tile goes global -> LDS with no VGPR destination and no
// ds_write.  A wave instruction writes 1 KiB linearly (M0 base + lane * 16), so the XOR swizzles of the two LDS
// layouts above are applied to the SOURCE address instead: lane -> physical 16-byte slot -> the logical vector
// that slot must hold.  Rows / columns past the operand edge are clamped onto valid memory (they only feed
// output rows / columns that the epilogue never stores); the K range must be whole 64-deep tiles.
template <int MODE, int EXT, int NT = 256>
struct GldsPlan {
  static constexpr int NV = EXT * BK / 8 / NT;   // 16-byte vectors per thread == 1-KiB pieces per wave
  const bf16_t* src[NV];                          // source of vector i at k = 0 of the operand
  long kstep;                                     // elements per unit of k
  __device__ __forceinline__ void init(const bf16_t* base, long ld, int r0, int rmax, int tid) {
    kstep = MODE == KC ? 1 : ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NT;                 // physical slot: LDS byte offset e * 16
      if (MODE == KC) {
        const int r = e >> 3, c = (e & 7) ^ (r & 7);
        const int gr = min(r0 + r, rmax - 1);
        src[i] = base + (long)gr * ld + c * 8;
      } else {
        const int kq = e / (EXT / 8), cbp = (e % (EXT / 8)) * 16;
        const int cbl = ((((cbp >> 5) ^ ks_swz<EXT>(kq)) << 5) | (cbp & 16));
        // whole 16-byte column groups: an extent that is not a multiple of 8 is rounded up into the row padding (the
        // launcher checks ld covers it); those columns only feed output rows / columns that are never stored
        const int rm8 = min((rmax + 7) & ~7, (int)ld);
        const int gc = min(r0 + cbl / 2, rm8 - 8);
        src[i] = base + (long)kq * ld + gc;
      }
    }
  }
  __device__ __forceinline__ void issue(char* lds, int k0, int tid) const {
    const int wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      typedef __attribute__((address_space(1))) const void* gptr;
      typedef __attribute__((address_space(3))) void* lptr;
      __builtin_amdgcn_global_load_lds((gptr)(src[i] + (long)k0 * kstep), (lptr)(lds + (i * NT + wv * 64) * 16), 16, 0, 0);
    }
  }
  // The same as opaque inline asm: the compiler orders every LDS access it can see behind ALL outstanding builtin DMA
  // loads; issued this way a stage can stay in flight across LDS reads and writes of other regions.  Completion is the
  // caller's to wait for (s_waitcnt vmcnt) before the barrier that publishes the stage.
  __device__ __forceinline__ void issue_asm(char* lds, int k0, int tid) const {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t wvo = (uint32_t)__builtin_amdgcn_readfirstlane((tid >> 6) * 64 * 16);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bf16_t* a = src[i] + (long)k0 * kstep;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(base + wvo + (uint32_t)(i * NT * 16)) : "memory");
    }
  }
};

// fragment of a 16-wide block `blk` (rows for KC, cols for KS) at k-step ks (32 k) of the staged tile
template <int MODE, int EXT>
__device__ __forceinline__ bf16x8 frag(const char* lds, int blk, int ks, int lane) {
  if (MODE == KC) {
    const int r = blk * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((c ^ (r & 7)) << 4));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int k0 = ks * 32 + g * 8;
    const int col = blk * 16 + pp * 4;
    typedef s4v __attribute__((address_space(3))) * lptr;
    (void)col;
    const int kl = k0 + q, kh = k0 + 4 + q;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + kl * (EXT * 2) + ((blk ^ ks_swz<EXT>(kl)) << 5) + pp * 8));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + kh * (EXT * 2) + ((blk ^ ks_swz<EXT>(kh)) << 5) + pp * 8));
    typedef short s8v __attribute__((ext_vector_type(8)));
    s8v t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, t);
  }
}

// K-slow fragments with OPAQUE transposing reads.  The compiler orders every LDS access it can see through the
// ds_read_tr builtin behind ALL outstanding LDS-DMA loads (it emitted s_waitcnt vmcnt(0) in front of the first
// ds_read_b64_tr_b16 of every K step), which serialised the prefetch of stage t+1 with the multiply of stage t in
// every <KC, KS> (data-gradient) GEMM.  As inline asm the reads carry no memory operand; the price is that their
// completion is ours to wait for: all the reads of one staged tile (both 32-deep k steps) are issued, then one
// s_waitcnt lgkmcnt(0), then empty asm statements that pin every consumer behind that wait.
template <int OFF>
__device__ __forceinline__ unsigned long long ds_read_tr16_b64(uint32_t a) {
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The swizzle code of a lane's k row does not change with the 32-deep k step nor between its low and high 4-row
// halves (ks_swz reads bits 0, 1 and 3 of k; those offsets move bits 2 and 5), so a lane has ONE address per 16-wide
// block and every read of a staged tile is that address plus an instruction immediate.
template <int EXT, int NB>
struct KsFrags {
  unsigned long long lo[BK / 32][NB], hi[BK / 32][NB];
  uint32_t addr[NB];
  __device__ __forceinline__ void init(const char* lds, int blk0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds;
    const int kl = g * 8 + q;
#pragma unroll
    for (int b = 0; b < NB; ++b) addr[b] = base + kl * (EXT * 2) + (((blk0 + b) ^ ks_swz<EXT>(kl)) << 5) + pp * 8;
  }
  // the tile staged `off` bytes behind the one init() was given
  __device__ __forceinline__ void read(uint32_t off) {
    static_for<BK / 32>([&](auto ks) {
      static_for<NB>([&](auto b) {
        const uint32_t a = addr[b] + off;
        lo[ks][b] = ds_read_tr16_b64<ks * 32 * EXT * 2>(a);
        hi[ks][b] = ds_read_tr16_b64<ks * 32 * EXT * 2 + 4 * EXT * 2>(a);
      });
    });
  }
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pin();
  }
  // every consumer of the fragments behind the s_waitcnt that precedes this in program order
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        asm volatile("" : "+v"(lo[ks][b]));
        asm volatile("" : "+v"(hi[ks][b]));
      }
  }
  __device__ __forceinline__ bf16x8 get(int ks, int b) const {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 t = {lo[ks][b], hi[ks][b]};
    return __builtin_bit_cast(bf16x8, t);
  }
};

// Second GEMM phase of a fused kernel: C2 (BMT rows of this workgroup, N2 columns) = T @ W2, T = the workgroup's BMT x 192
// bf16 tile in LDS (rows RSB bytes apart), W2 (192, N2) row-major (K-slow), four waves as 2 x 2 of (BMT / 2) x 64, 128
// columns of C2 at a time, the W2 panel staged by LDS-DMA through two 16 KiB buffers.
// WMODE = KS: W2 (192, N2) row-major (a weight as stored, used as a data gradient); WMODE = KC: W2 (N2, 192) row-major
// (a weight as stored, used forward: C2 = T @ W2^T).
// rowof(tile row) = row of C2, or -1 for a tile row that is not stored (the pooling-row tiles of convpool_dgrad.hip map
// their rows to scattered memory tokens; the GEMM kernels' tiles are runs of consecutive rows).
// PIPE: the W2 stages are issued as opaque DMA (GldsPlan::issue_asm) and the first stage of the NEXT 128-column panel is
// requested before this panel's epilogue, so it arrives under the slab round trip and the C2 stores instead of costing an
// exposed L2 round trip per panel (three per launch at N2 = 384); same values.
template <int BMT, int RSB, int WMODE, class RowOf, bool PIPE = false>
__device__ __forceinline__ void tile_times_w2_rows(const char* tile, char* ldsB, char* slabs, const bf16_t* W2, long ldw2, int N2,
                                                   bf16_t* C2, RowOf rowof, int tid) {
  constexpr int MB2 = BMT / 32, NB2 = 4, K2 = 192, STG = 128 * BK * 2, NKT = K2 / BK;
  const int lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  KsFrags<128, WMODE == KS ? NB2 : 1> kb;
  if constexpr (WMODE == KS) kb.init(ldsB, wn * NB2, lane);
  constexpr int RS = 64 * 2 + 16, CH = 64 / 8;            // epilogue slab: 32 rows x 64 columns per wave
  char* my = slabs + wv * (32 * RS);
  int par = 0;                                            // PIPE: ring buffer of the next stage to be consumed
  if constexpr (PIPE) {
    GldsPlan<WMODE, 128, 256> g0;
    g0.init(W2, ldw2, 0, N2, tid);
    g0.issue_asm(ldsB, 0, tid);
  }
  for (int n0 = 0; n0 < N2; n0 += 128) {
    GldsPlan<WMODE, 128, 256> gb;
    gb.init(W2, ldw2, n0, N2, tid);
    f32x4 acc[NB2][MB2];
#pragma unroll
    for (int a = 0; a < NB2; ++a)
#pragma unroll
      for (int b = 0; b < MB2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else gb.issue(ldsB, 0, tid);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int cur = PIPE ? par : (kt & 1);
      if constexpr (PIPE) {
        if (kt + 1 < NKT) gb.issue_asm(ldsB + (cur ^ 1) * STG, (kt + 1) * BK, tid);
        else if (n0 + 128 < N2) {
          GldsPlan<WMODE, 128, 256> gn;
          gn.init(W2, ldw2, n0 + 128, N2, tid);
          gn.issue_asm(ldsB + (cur ^ 1) * STG, 0, tid);
        }
        par = cur ^ 1;
      } else if (kt + 1 < NKT) gb.issue(ldsB + (cur ^ 1) * STG, (kt + 1) * BK, tid);
      if constexpr (WMODE == KS) {
        kb.read(cur * STG);
        kb.wait();
      }
#pragma unroll
      for (int ks = 0; ks < BK / 32; ++ks) {
        bf16x8 fa[MB2], fb[NB2];
#pragma unroll
        for (int b = 0; b < MB2; ++b)
          fa[b] = *reinterpret_cast<const bf16x8*>(tile + (wm * (BMT / 2) + b * 16 + (lane & 15)) * RSB +
                                                   (kt * BK + ks * 32 + (lane >> 4) * 8) * 2);
#pragma unroll
        for (int a = 0; a < NB2; ++a) {
          if constexpr (WMODE == KS) fb[a] = kb.get(ks, a);
          else fb[a] = frag<KC, 128>(ldsB + cur * STG, wn * NB2 + a, ks, lane);
        }
#pragma unroll
        for (int a = 0; a < NB2; ++a)
#pragma unroll
          for (int b = 0; b < MB2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[a], fa[b], acc[a][b], 0, 0, 0);
      }
      if constexpr (PIPE) {
        // the next stage of THIS panel is needed right away; the next panel's first stage is waited for at the top of
        // its loop (the buffer this k tile read is overwritten only behind that barrier)
        if (kt + 1 < NKT) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      } else __syncthreads();
    }
    // bf16 through the wave's slab: 16-byte stores, whole 128-byte row segments
#pragma unroll
    for (int h = 0; h < (MB2 + 1) / 2; ++h) {
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int b = 2 * h + bb;
        if (b < MB2) {
#pragma unroll
          for (int a = 0; a < NB2; ++a) {
            const f32x4 v = acc[a][b];
            uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<uint2*>(my + (bb * 16 + (lane & 15)) * RS + (a * 16 + (lane >> 4) * 4) * 2) = pk;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      constexpr int ROWS = MB2 >= 2 ? 32 : 16;
#pragma unroll
      for (int i = 0; i < ROWS * CH / 64; ++i) {
        const int idx = i * 64 + lane, r = idx / CH, ch = idx - r * CH;
        const int m = rowof(wm * (BMT / 2) + h * 32 + r), n = n0 + wn * 64 + ch * 8;
        if (m >= 0 && n < N2) *reinterpret_cast<u32x4*>(C2 + (long)m * N2 + n) = *reinterpret_cast<const u32x4*>(my + r * RS + ch * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int BMT, int RSB, int WMODE>
__device__ __forceinline__ void tile_times_w2(const char* tile, char* ldsB, char* slabs, const bf16_t* W2, long ldw2, int N2,
                                              bf16_t* C2, int m0, int M, int tid) {
  tile_times_w2_rows<BMT, RSB, WMODE>(tile, ldsB, slabs, W2, ldw2, N2, C2,
                                      [m0, M](int r) { return m0 + r < M ? m0 + r : -1; }, tid);
}

}  // namespace
