// hipcc-flags: -fgpu-flush-denormals-to-zero
// Adjoint of the x_proj linear layer of both scan directions, fused with the sum over the scan
// backward's channel-chunk partials.  Replaces, per block and per step, the einsum / addmm chain of
// selective_scan_interface.py:698-734:
//     dx_dbl = sum_chunks dx_dbl_partial                              (was fv_reduce_partials)
//     d x_proj.weight[dir] += dx_dbl[dir]^T @ xc[dir]                 (was a cast + baddbmm_)
//     dxc[dir] = dxc_scan[dir] + dx_dbl[dir] @ x_proj.weight[dir]     (was a copy + baddbmm)
// One lane per channel d: it keeps column d of the weight (W values) and W accumulators of the
// weight gradient in registers; the dx_dbl rows of the block's row slice are LDS broadcasts.  Blocks
// emit per-slice partials of the weight gradient for the fixed-order reduction (no atomics).
#include <stdlib.h>

#include "rowwalk.h"

namespace {

struct XprojParams {
  const float* dxdbl_part;   // (nchunks, 2, M, W) fp32
  const void* xc;            // (2, M, d_in)
  const float* Wx[2];        // (W, d_in) fp32
  float* dxc;                // (2, M, d_in) fp32: in = through-the-scan part, out = total
  float* dW_part;            // (nslices, 2, W, d_in) fp32 (null: the weight gradient is computed elsewhere)
  void* dxdbl_out;           // (2, M, WP) bf16, WP = W rounded up to 8, pad columns zero (nullable): summed dx_dbl
  const void* Wxt;           // (2, d_in, W) bf16: the TRANSPOSED compute-dtype shadow of both weights (xproj_bwd_mmb_kernel)
  void* dxc2_out;            // (2, M, d_in) bf16 (nullable): the product is WRITTEN here instead of added to dxc
  int nchunks, M, d_in, rows_per_block;
};

template <typename T, int W, int RB, bool DW>
__global__ __launch_bounds__(256) void xproj_bwd_kernel(XprojParams p) {
  extern __shared__ __attribute__((aligned(16))) float s_dx[];     // rows_per_block * W
  const int dir = blockIdx.z, slice = blockIdx.y;
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = d < p.d_in;
  const int dd = act ? d : 0;
  const int m0 = slice * p.rows_per_block;
  const int nr = min(p.rows_per_block, p.M - m0);
  // this lane's column of W_x: needed after the barrier, requested before the staging loop so that it arrives under it
  float wcol[W], acc[W];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    wcol[c] = p.Wx[dir][(size_t)c * p.d_in + dd];
    acc[c] = 0.f;
  }
  // stage the slice's dx_dbl rows, summing the channel-chunk partials in fixed order.  The block is short-lived and
  // all latency: six partials of two elements are in flight per lane (a chunk loop with a run-time bound issues one
  // load, waits, adds, and repeats)
  for (int e = threadIdx.x; e < nr * W; e += 2 * blockDim.x) {
    const int e1 = e + blockDim.x;
    const bool two = e1 < nr * W;
    float t0 = 0.f, t1 = 0.f;
    for (int c0 = 0; c0 < p.nchunks; c0 += 6) {
      float v0[6], v1[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const bool on = c0 + u < p.nchunks;
        const size_t o = (((size_t)(on ? c0 + u : 0) * 2 + dir) * p.M + m0) * W;
        v0[u] = on ? p.dxdbl_part[o + e] : 0.f;
        v1[u] = on && two ? p.dxdbl_part[o + e1] : 0.f;
      }
      t0 += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + (v0[4] + v0[5]);
      t1 += ((v1[0] + v1[1]) + (v1[2] + v1[3])) + (v1[4] + v1[5]);
    }
    s_dx[e] = t0;
    if (two) s_dx[e1] = t1;
  }
  __syncthreads();
  if (p.dxdbl_out && blockIdx.x == 0) {      // one channel block publishes the summed rows (bf16, padded row stride)
    constexpr int WP = (W + 7) / 8 * 8;
    bf16_t* o = (bf16_t*)p.dxdbl_out + ((size_t)dir * p.M + m0) * WP;
    for (int e = threadIdx.x; e < nr * WP; e += blockDim.x) {
      const int r = e / WP, c = e - r * WP;
      o[e] = __float2bfloat16(c < W ? s_dx[r * W + c] : 0.f);
    }
  }
  const T* xc = (const T*)p.xc + ((size_t)dir * p.M + m0) * p.d_in + dd;
  float* dxc = p.dxc + ((size_t)dir * p.M + m0) * p.d_in + dd;
  for (int r0 = 0; r0 < nr; r0 += RB) {       // RB rows of loads in flight per lane
    float xv[RB], base[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int r = min(r0 + k, nr - 1);
      xv[k] = io<T>::ld(xc + (size_t)r * p.d_in);
      base[k] = dxc[(size_t)r * p.d_in];
    }
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      if (r0 + k < nr) {
        const float* row = s_dx + (r0 + k) * W;
        float o = base[k];
#pragma unroll
        for (int c = 0; c < W; ++c) {
          const float g = row[c];
          o = fmaf(g, wcol[c], o);
          if constexpr (DW) acc[c] = fmaf(g, xv[k], acc[c]);
        }
        if (act) dxc[(size_t)(r0 + k) * p.d_in] = o;
      }
    }
  }
  if (DW && act) {
    float* dst = p.dW_part + (((size_t)slice * 2 + dir) * W) * p.d_in + d;
#pragma unroll
    for (int c = 0; c < W; ++c) dst[(size_t)c * p.d_in] = acc[c];
  }
}

// ---- the data half of the adjoint on the fp32 matrix cores (round 6; taken when the weight gradient is computed elsewhere
// -- the grouped launch of the flat training state -- i.e. DW = false above).  The lane-per-channel kernel spends W FMAs
// and W / 4 LDS broadcast reads per (row, channel) on the vector pipe: 37 us at FastVim-B (M = 3 584 pooled rows, 1 536
// channels, W = 80), 49 us on the un-pooled Vim-T, for a product of 0.9 GFLOP.  v_mfma_f32_16x16x4_f32 is an exact fp32 FMA
// chain (the scan kernels' dt_proj uses it the same way), so the values keep fp32 operands and fp32 accumulation:
//     dxc[m][d] += sum_w G[m][w] Wx[w][d],   G = sum over the channel-chunk partials of d x_dbl (fixed order)
// A workgroup owns 64 rows: G is staged once in LDS (row stride = 4 (mod 64) words x odd: the 16 x 4 operand reads of a k
// step hit 64 banks); its four waves own 64 channels each of a 256-channel block and walk `cbw` such blocks; the weight
// operand needs no transposition in this layout -- lane (k = lane >> 4, n = lane & 15) reads Wx[k][n], 64 contiguous bytes per
// k row -- and comes straight from L2.  The weight rides in the A slot, so a lane ends with four consecutive channels of one
// row: d xc is read and written 16 bytes per lane.
typedef float xm_f32x4 __attribute__((ext_vector_type(4)));

template <int W>
__global__ __launch_bounds__(256, 2) void xproj_bwd_mm_kernel(XprojParams p, int cbw) {
  constexpr int KQ = (W + 3) / 4, W4 = 4 * KQ, WS = (W4 / 4) % 2 ? W4 : W4 + 4;
  extern __shared__ __attribute__((aligned(16))) float s_g[];      // 64 * WS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int dir = blockIdx.z, m0 = blockIdx.y * 64;
  const int nr = min(64, p.M - m0);
  // stage the rows, summing the channel-chunk partials in fixed order (six in flight per lane, like the kernel above)
  if (W % 4 == 0 && p.nchunks == 1) {
    // already summed: every 16-byte piece of the 64 rows requested at once (see xproj_bwd_mmb_kernel: the general loop has
    // one load in flight per lane and trip with a single chunk)
    constexpr int NV = (64 * (W4 / 4) + 255) / 256;
    const float* src = p.dxdbl_part + ((size_t)dir * p.M + m0) * W;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * 256, r = e / (W4 / 4), c4 = e - r * (W4 / 4);
      const bool ok = e < 64 * (W4 / 4) && r < nr && 4 * c4 < W;
      v[i] = *reinterpret_cast<const float4*>(src + (ok ? r * W + 4 * c4 : 0));
      if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * 256, r = e / (W4 / 4), c4 = e - r * (W4 / 4);
      if (e < 64 * (W4 / 4)) *reinterpret_cast<float4*>(s_g + r * WS + 4 * c4) = v[i];
    }
  } else {
  {
#pragma clang fp reassociate(off) contract(off)      // (the published rows are the same bits in every form of this kernel)
  for (int e = tid; e < 64 * W4; e += 256) {
    const int r = e / W4, c = e - r * W4;
    float t = 0.f;
    if (r < nr && c < W) {
      for (int c0 = 0; c0 < p.nchunks; c0 += 6) {
        float v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const bool on = c0 + u < p.nchunks;
          v[u] = on ? p.dxdbl_part[(((size_t)(c0 + u) * 2 + dir) * p.M + m0 + r) * W + c] : 0.f;
        }
        t += ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
      }
    }
    s_g[r * WS + c] = t;
  }
  }
  }
  __syncthreads();
  if (p.dxdbl_out && blockIdx.x == 0) {      // one channel block publishes the summed rows (bf16, padded row stride)
    constexpr int WP = (W + 7) / 8 * 8;
    bf16_t* o = (bf16_t*)p.dxdbl_out + ((size_t)dir * p.M + m0) * WP;
    for (int e = tid; e < nr * WP; e += 256) {
      const int r = e / WP, c = e - r * WP;
      o[e] = __float2bfloat16(c < W ? s_g[r * WS + c] : 0.f);
    }
  }
  const int mr = lane & 15, kq = lane >> 4;
  // Buffer addressing (rowwalk.h): 32-bit lane offsets -- with flat pointers every one of the 144 loads of a channel block
  // carried its own 64-bit address pair and the kernel spilled.  The whole offset is in the LANE part (the range check
  // does not see a scalar offset): rows past the slice's end and weight rows past W lie beyond their descriptors, so loads
  // return zero and stores are dropped -- no branch, no clamp.
  const __amdgpu_buffer_rsrc_t bw_ = fv_make_buf(p.Wx[dir], (size_t)W * p.d_in * 4);
  const __amdgpu_buffer_rsrc_t bd_ = fv_make_buf(p.dxc + ((size_t)dir * p.M + m0) * p.d_in, (size_t)nr * p.d_in * 4);
  const int row4 = 4 * p.d_in * 4;              // bytes between k steps of the weight / row quads of d xc
  for (int cb = 0; cb < cbw; ++cb) {
    const int n0 = (blockIdx.x * cbw + cb) * 256 + wv * 64;
    if (n0 >= p.d_in) break;                  // uniform per wave (d_in is a multiple of 64: a wave's 64 channels are all live)
    xm_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (xm_f32x4){0.f, 0.f, 0.f, 0.f};
    int wof[4], dof[4];                       // weight: row kq of a k step, channel n0 + 16 b + mr; d xc: row mr of a row block, channels n0 + 16 b + 4 kq ..
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      wof[b] = (kq * p.d_in + n0 + b * 16 + mr) * 4;
      dof[b] = (mr * p.d_in + n0 + b * 16 + 4 * kq) * 4;
    }
    // EVERY weight value of this channel block is requested before the first MFMA (left to the compiler, a k step's four
    // loads were issued one step ahead and waited for in place: an L2 round trip per k step)
    float bw[KQ][4];
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t t[1];
        fv_buf_load_words<1>(bw_, wof[b] + ks * row4, 0, t);
        bw[ks][b] = __uint_as_float(t[0]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KQ; ++ks) {
      float av[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = s_g[(a * 16 + mr) * WS + 4 * ks + kq];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[ks][b], av[a], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // acc[a][b][j] = product[row 16 a + mr][channel n0 + 16 b + 4 kq + j] (the weight rides in the A slot, as in
    // gemm_mfma.hip: a lane ends with four consecutive channels of one row -- 16-byte accesses).  d xc is read and written a
    // row block at a time, its four loads in flight together: as `*q += v` through one pointer the read-modify-writes are
    // a chain of dependent HBM round trips to the compiler, which cannot tell the addresses apart
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      uint32_t base[4][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) fv_buf_load_words<4>(bd_, dof[b] + a * 16 * (p.d_in * 4), 0, base[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(__uint_as_float(base[b][j]) + acc[a][b][j]);
        fv_buf_store_words<4>(bd_, dof[b] + a * 16 * (p.d_in * 4), 0, o);
      }
    }
  }
}

// The same data half on the BF16 matrix cores (round 6): bf16(summed d x_dbl) times the bf16 shadow weight -- the operands the
// reference's autocast backward multiplies (selective_scan_interface.py:726-734), fp32 accumulate, fp32 d xc.  The fp32 form
// above spends 4.3 us of matrix time per 256-channel block (20 k steps x 16 `v_mfma_f32_16x16x4_f32` per wave); here a
// block is 3 k steps x 16 `v_mfma_f32_16x16x32_bf16`, and the launch sits on the round trip of d xc.  The weight rides in
// the A slot (a lane ends with four consecutive channels of one row: 16-byte read-modify-writes), so a lane needs eight
// consecutive k of ONE channel per fragment: the weight comes TRANSPOSED, (d_in, W) bf16 -- one 16-byte load per
// fragment -- from the shadow the flat training state re-makes after every optimizer step.  W % 8 == 0; a row of the
// transposed weight is W values, the k padding up to 32 KS reads on into the next channel's row (finite; it meets the
// zero padding of the d x_dbl rows) or beyond the descriptor (zero).
typedef __bf16 xm_bf16x8 __attribute__((ext_vector_type(8)));
template <int W>
__global__ __launch_bounds__(256, 2) void xproj_bwd_mmb_kernel(XprojParams p, int cbw) {
  static_assert(W % 8 == 0, "transposed weight rows must be 16-byte multiples");
  constexpr int KS = (W + 31) / 32, KP = 32 * KS, WS = KP + 4;      // row stride of the staged rows: KP + 4 floats
  extern __shared__ __attribute__((aligned(16))) float s_g[];      // 64 * WS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int dir = blockIdx.z, m0 = blockIdx.y * 64;
  const int nr = min(64, p.M - m0);
  const int mr = lane & 15, kq = lane >> 4;
  // the weight fragments of the workgroup's FIRST channel block (usually its only one) are requested before the rows
  // are staged: the block is a chain of dependent round trips -- rows, barrier, weights, product, store -- and this
  // takes the weights off it
  const __amdgpu_buffer_rsrc_t bw_ = fv_make_buf((const bf16_t*)p.Wxt + (size_t)dir * p.d_in * W, (size_t)p.d_in * W * 2);
  auto load_aw = [&](int n0, xm_bf16x8 (&aw)[4][KS]) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint32_t t[4];
        fv_buf_load_words<4>(bw_, ((n0 + b * 16 + mr) * W + 32 * ks + 8 * kq) * 2, 0, t);
        aw[b][ks] = __builtin_bit_cast(xm_bf16x8, *reinterpret_cast<fv_u32x4*>(t));
      }
  };
  xm_bf16x8 aw[4][KS];
  load_aw(blockIdx.x * cbw * 256 + wv * 64, aw);
  // stage the rows, summing the channel-chunk partials in fixed order; columns W .. KP and rows past the slice zero
  if (p.nchunks == 1) {
    // already summed (the wide models pre-sum their eight chunks): every 16-byte piece of the 64 rows requested at once
    // -- the general loop below has ONE load in flight per lane and iteration with a single chunk, twenty-four dependent
    // round trips in front of the barrier (round 6: that loop was most of the kernel's 20 us)
    constexpr int NV = (64 * (KP / 4) + 255) / 256;
    const float* src = p.dxdbl_part + ((size_t)dir * p.M + m0) * W;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * 256, r = e / (KP / 4), c4 = e - r * (KP / 4);
      const bool ok = e < 64 * (KP / 4) && r < nr && 4 * c4 < W;
      v[i] = *reinterpret_cast<const float4*>(src + (ok ? r * W + 4 * c4 : 0));
      if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * 256, r = e / (KP / 4), c4 = e - r * (KP / 4);
      if (e < 64 * (KP / 4)) *reinterpret_cast<float4*>(s_g + r * WS + 4 * c4) = v[i];
    }
  } else {
    // four elements per lane and trip, each with up to six chunk partials in flight (one element per trip left a single
    // dependent round trip per chunk group and trip: sixteen trips at the channel model's four chunks)
#pragma clang fp reassociate(off) contract(off)      // (the published rows are the same bits in every form of this kernel)
    for (int e0 = tid; e0 < 64 * KP; e0 += 4 * 256) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      int off[4];
      bool ok[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 256, r = e / KP, c = e - r * KP;
        ok[j] = e < 64 * KP && r < nr && c < W;
        off[j] = ok[j] ? r * W + c : 0;
      }
      for (int c0 = 0; c0 < p.nchunks; c0 += 6) {
        float v[4][6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const bool on = c0 + u < p.nchunks;
          const float* src = p.dxdbl_part + (((size_t)(on ? c0 + u : 0) * 2 + dir) * p.M + m0) * W;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = src[off[j]];
            v[j][u] = on ? x : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] += ((v[j][0] + v[j][1]) + (v[j][2] + v[j][3])) + (v[j][4] + v[j][5]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 256;
        if (e < 64 * KP) {
          const int r = e / KP, c = e - r * KP;
          s_g[r * WS + c] = ok[j] ? t[j] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (p.dxdbl_out && blockIdx.x == 0) {      // one channel block publishes the summed rows (bf16; W is a multiple of 8)
    bf16_t* o = (bf16_t*)p.dxdbl_out + ((size_t)dir * p.M + m0) * W;
    for (int e = tid; e < nr * W; e += 256) {
      const int r = e / W, c = e - r * W;
      o[e] = __float2bfloat16(s_g[r * WS + c]);
    }
  }
  // this lane's B operands, the same for every channel block: rows 16 a + mr, k = 32 ks + 8 kq .. + 7 (rounded to bf16 as
  // the published rows are)
  xm_bf16x8 bv[4][KS];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* q = s_g + (a * 16 + mr) * WS + 32 * ks + 8 * kq;
      const float4 lo = *reinterpret_cast<const float4*>(q), hi = *reinterpret_cast<const float4*>(q + 4);
      uint32_t w4[4] = {pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
      bv[a][ks] = __builtin_bit_cast(xm_bf16x8, *reinterpret_cast<fv_u32x4*>(w4));
    }
  const __amdgpu_buffer_rsrc_t bd_ = fv_make_buf(p.dxc + ((size_t)dir * p.M + m0) * p.d_in, (size_t)nr * p.d_in * 4);
  const __amdgpu_buffer_rsrc_t b2_ = fv_make_buf((bf16_t*)p.dxc2_out + ((size_t)dir * p.M + m0) * p.d_in, p.dxc2_out ? (size_t)nr * p.d_in * 2 : 0);
  for (int cb = 0; cb < cbw; ++cb) {
    const int n0 = (blockIdx.x * cbw + cb) * 256 + wv * 64;
    if (n0 >= p.d_in) break;                  // uniform per wave (d_in is a multiple of 64)
    xm_f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (xm_f32x4){0.f, 0.f, 0.f, 0.f};
    int dof[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) dof[b] = (mr * p.d_in + n0 + b * 16 + 4 * kq) * 4;
    if (cb > 0) load_aw(n0, aw);              // (further blocks: every fragment requested before the first MFMA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[b][ks], bv[a][ks], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // acc[a][b][j] = product[row 16 a + mr][channel n0 + 16 b + 4 kq + j]
    if (p.dxc2_out) {
      // the product leaves as its own bf16 tensor -- the conv + pool adjoint takes the pooled gradient as two addends
      // (fv_mixer_conv_pool_bwd2) -- instead of a read-modify-write of the fp32 d xc: 2 bytes per element instead of 8
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t o[2] = {pack_bf16x2(acc[a][b][0], acc[a][b][1]), pack_bf16x2(acc[a][b][2], acc[a][b][3])};
          fv_buf_store_words<2>(b2_, (dof[b] + a * 16 * (p.d_in * 4)) >> 1, 0, o);
        }
      continue;
    }
    // d xc is read and written a row block at a time
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      uint32_t base[4][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) fv_buf_load_words<4>(bd_, dof[b] + a * 16 * (p.d_in * 4), 0, base[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __float_as_uint(__uint_as_float(base[b][j]) + acc[a][b][j]);
        fv_buf_store_words<4>(bd_, dof[b] + a * 16 * (p.d_in * 4), 0, o);
      }
    }
  }
}

// ---- x_proj forward: x_dbl[dir] = xc[dir] @ Wx[dir]^T, a (B*Lc, d_in) x (d_in, R+2N <= 112) product per direction
// (mamba_simple_faster.py:321-327).  It is tiny (0.12 GFLOP at FastVim-T) and sits on the critical path between the
// conv and the scan: one wave per 16 rows, operands straight from global memory into MFMA fragments (no LDS, no
// barrier), every load of a row block in flight before the first MFMA.  hipBLASLt took 7.3 us for it.
typedef __bf16 xp_bf16x8 __attribute__((ext_vector_type(8)));
typedef float xp_f32x4 __attribute__((ext_vector_type(4)));

template <int NBK, int NWV>      // 16-column blocks of the output (NBK * 16 >= width); waves that split K
__global__ __launch_bounds__(64 * NWV) void xproj_fwd_kernel(const bf16_t* __restrict__ xc, const bf16_t* __restrict__ Wx,
                                                        bf16_t* __restrict__ out, int M, int K, int W) {
  // NWV waves split K (each keeps its share of loads in flight), then sum through LDS in a fixed order
  __shared__ xp_f32x4 s_acc[NWV - 1][NBK][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, kc = lane >> 4;
  const int dir = blockIdx.y, m0 = blockIdx.x * 16;
  const int m = min(m0 + r, M - 1);
  const int ksteps = K / 32, per = (ksteps + NWV - 1) / NWV;
  const int k_lo = wv * per * 32, k_hi = min(K, (wv + 1) * per * 32);
  const bf16_t* a_row = xc + ((size_t)dir * M + m) * K + kc * 8;
  const bf16_t* b_row[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) b_row[nb] = Wx + ((size_t)dir * W + min(nb * 16 + r, W - 1)) * K + kc * 8;
  xp_f32x4 acc[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) acc[nb] = (xp_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
  for (int k = k_lo; k < k_hi; k += 32) {
    const xp_bf16x8 a = *reinterpret_cast<const xp_bf16x8*>(a_row + k);
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      const xp_bf16x8 b = *reinterpret_cast<const xp_bf16x8*>(b_row[nb] + k);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[nb], 0, 0, 0);   // rows = n, cols = m
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) s_acc[wv - 1][nb][lane] = acc[nb];
  }
  __syncthreads();
  // acc[nb][j] = C[m0 + (lane & 15)][nb * 16 + (lane >> 4) * 4 + j]
  if (wv == 0 && m0 + r < M) {
    bf16_t* o = out + ((size_t)dir * M + m0 + r) * W;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      xp_f32x4 t = acc[nb];
#pragma unroll
      for (int w = 0; w < NWV - 1; ++w) t += s_acc[w][nb][lane];
      const int n = nb * 16 + kc * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < W) o[n + j] = __float2bfloat16(t[j]);
    }
  }
}

// ---- summed d x_dbl rows as the bf16 operand of the x_proj weight gradient, for up to 64 mixers in ONE launch: where
// the x_proj adjoint's data half runs inside the scan backward (fv_mixer_scan_bwd_xproj), nothing else has summed the
// channel-chunk partials.  out[row][c] = bf16(sum_chunk in[chunk][row][c]) for c < W, zero in the pad columns (the
// rows fv_mixer_xproj_bwd2 publishes, bit for bit: same summation order).
constexpr int CRJ_MAX = 64;
struct ChunkRowJobs {
  const float* in[CRJ_MAX];
  bf16_t* out[CRJ_MAX];
  int nchunks, W, WP;
  long rows;
};
__global__ __launch_bounds__(256) void chunk_rows_bf16_kernel(ChunkRowJobs J) {
  const float* __restrict__ in = J.in[blockIdx.y];
  bf16_t* __restrict__ out = J.out[blockIdx.y];
  const long n = J.rows * J.WP;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const long r = e / J.WP;
    const int c = (int)(e - r * J.WP);
    float t = 0.f;
    if (c < J.W) {
      const float* src = in + r * J.W + c;
      const size_t cs = (size_t)J.rows * J.W;
      for (int c0 = 0; c0 < J.nchunks; c0 += 6) {
        float v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) v[u] = c0 + u < J.nchunks ? src[(size_t)(c0 + u) * cs] : 0.f;
        t += ((v[0] + v[1]) + (v[2] + v[3])) + (v[4] + v[5]);
      }
    }
    out[e] = __float2bfloat16(t);
  }
}

}  // namespace

extern "C" int fv_chunk_rows_bf16(const float* const* partials, void* const* outs, int njobs, int nchunks, long rows,
                                  int width, fv_stream_t stream) {
  FV_CHECK(partials && outs && njobs > 0 && njobs <= CRJ_MAX, "chunk_rows_bf16: 1..%d jobs", CRJ_MAX);
  FV_CHECK(nchunks > 0 && rows > 0 && width > 0, "chunk_rows_bf16: empty dimension");
  ChunkRowJobs J{};
  for (int j = 0; j < njobs; ++j) {
    FV_CHECK(partials[j] && outs[j], "chunk_rows_bf16: null pointer in job %d", j);
    J.in[j] = partials[j];
    J.out[j] = (bf16_t*)outs[j];
  }
  J.nchunks = nchunks; J.W = width; J.WP = (width + 7) / 8 * 8; J.rows = rows;
  const int bx = fv_cdiv(rows * J.WP, 256 * 4) < 1 ? 1 : fv_cdiv(rows * J.WP, 256 * 4);
  hipLaunchKernelGGL(chunk_rows_bf16_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, J);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

static int xproj_rows() {
  static const int r = fv_tune("FASTVIM_XPROJ_ROWS", 16);   // tuning hook
  return r;
}
extern "C" int fv_mixer_xproj_bwd_slices(int M) { return fv_cdiv(M, xproj_rows()); }

extern "C" int fv_mixer_xproj_bwd(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                                  const float* x_proj_w_b, float* dxc, float* dW_partials, int M, int d_inner,
                                  int width, int dtype, fv_stream_t stream) {
  return fv_mixer_xproj_bwd2(dx_dbl_partials, nchunks, xc, x_proj_w, x_proj_w_b, dxc, dW_partials, nullptr, M, d_inner,
                             width, dtype, stream);
}

// the data half with the transposed bf16 shadow weights (2, d_inner, width): bf16 matrix cores where the matrix-core form
// applies (fv_mixer_xproj_bwd3_ok), else fv_mixer_xproj_bwd2
extern "C" int fv_mixer_xproj_bwd3_ok(int M, int d_inner, int width, int dtype) {
  static const bool on = (fv_tune("FASTVIM_XPROJ_BWD_MMB", 1) != 0);   // tuning hook
  const bool built = width == 56 || width == 80 || width == 96 || width == 112 || width == 64;
  return on && built && dtype == FV_BF16 && d_inner % 64 == 0 && d_inner >= 768 && M > 0 &&
         (size_t)M * d_inner * 4 < 0x7fffffffull && (size_t)d_inner * width * 2 < 0x7fffffffull;
}
extern "C" int fv_mixer_xproj_bwd3(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                                   const float* x_proj_w_b, const void* x_proj_w_t_bf16, float* dxc, void* dxc2_bf16,
                                   void* dx_dbl_bf16, int M, int d_inner, int width, int dtype, fv_stream_t stream) {
  const bool ok = x_proj_w_t_bf16 && dx_dbl_bf16 && fv_mixer_xproj_bwd3_ok(M, d_inner, width, dtype);
  if (!ok) {
    FV_CHECK(!dxc2_bf16, "mixer_xproj_bwd3: a separate bf16 product needs the bf16 matrix-core form (fv_mixer_xproj_bwd3_ok)");
    return fv_mixer_xproj_bwd2(dx_dbl_partials, nchunks, xc, x_proj_w, x_proj_w_b, dxc, nullptr, dx_dbl_bf16, M, d_inner,
                               width, dtype, stream);
  }
  FV_CHECK(dx_dbl_partials && xc && (dxc || dxc2_bf16), "mixer_xproj_bwd3: null pointer");
  FV_CHECK(nchunks > 0, "mixer_xproj_bwd3: empty dimension");
  XprojParams p{};
  p.dxdbl_part = dx_dbl_partials; p.xc = xc; p.Wx[0] = x_proj_w; p.Wx[1] = x_proj_w_b; p.dxc = dxc; p.dxc2_out = dxc2_bf16;
  p.dxdbl_out = dx_dbl_bf16; p.Wxt = x_proj_w_t_bf16; p.nchunks = nchunks; p.M = M; p.d_in = d_inner;
  const int slices = fv_cdiv(M, 64), cblocks = fv_cdiv(d_inner, 256);
  int cbw = 1;
  while (cbw < cblocks && (long)slices * 2 * fv_cdiv(cblocks, cbw * 2) >= 2 * fv_cu_count()) cbw *= 2;
  const dim3 mgrid(fv_cdiv(cblocks, cbw), slices, 2);
#define FV_XB(WW)                                                                                              \
  hipLaunchKernelGGL((xproj_bwd_mmb_kernel<WW>), mgrid, dim3(256), (size_t)64 * (32 * ((WW + 31) / 32) + 4) * 4, \
                     (hipStream_t)stream, p, cbw)
  switch (width) {
    case 56: FV_XB(56); break;
    case 80: FV_XB(80); break;
    case 96: FV_XB(96); break;
    case 112: FV_XB(112); break;
    case 64: FV_XB(64); break;
    default: return FV_ERR_UNSUPPORTED;
  }
#undef FV_XB
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_mixer_xproj_bwd2(const float* dx_dbl_partials, int nchunks, const void* xc, const float* x_proj_w,
                                   const float* x_proj_w_b, float* dxc, float* dW_partials, void* dx_dbl_bf16, int M,
                                   int d_inner, int width, int dtype, fv_stream_t stream) {
  FV_CHECK(dx_dbl_partials && xc && x_proj_w && x_proj_w_b && dxc && (dW_partials || dx_dbl_bf16),
           "mixer_xproj_bwd: null pointer");
  FV_CHECK(M > 0 && d_inner > 0 && nchunks > 0, "mixer_xproj_bwd: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer_xproj_bwd: dtype must be fp32 or bf16");
  XprojParams p{};
  p.dxdbl_part = dx_dbl_partials; p.xc = xc; p.Wx[0] = x_proj_w; p.Wx[1] = x_proj_w_b; p.dxc = dxc;
  p.dW_part = dW_partials; p.dxdbl_out = dx_dbl_bf16; p.nchunks = nchunks; p.M = M; p.d_in = d_inner; p.rows_per_block = xproj_rows();
  static const bool mm = (fv_tune("FASTVIM_XPROJ_BWD_MM", 1) != 0);   // tuning hook
  // (same-box A/B, profiles/r06_ab_xproj_bwd_mm.log: FastVim-B 224 px -0.4 ... -0.6 %, channel model -0.25 %, 2048 px even; at
  //  d_inner 384 -- un-pooled Vim-T, 51 200 pooled rows -- the lane-per-channel kernel already sits on the 157 MB d xc
  //  round trip and the matrix-core form is 0.3-0.5 % slower: taken from d_inner 768 up)
  if (!dW_partials && mm && d_inner % 64 == 0 && d_inner >= 768 && (size_t)M * d_inner * 4 < 0x7fffffffull) {
    // the data half on the fp32 matrix cores: 64-row workgroups walking `cbw` blocks of 256 channels -- as many as keep
    // about two workgroups per CU's worth of them (fewer, longer workgroups re-stage the rows less often)
    const int slices = fv_cdiv(M, 64), cblocks = fv_cdiv(d_inner, 256);
    int cbw = 1;
    while (cbw < cblocks && (long)slices * 2 * fv_cdiv(cblocks, cbw * 2) >= 2 * fv_cu_count()) cbw *= 2;
    const dim3 mgrid(fv_cdiv(cblocks, cbw), slices, 2);
    hipStream_t mst = (hipStream_t)stream;
#define FV_XM(WW)                                                                                              \
  do {                                                                                                         \
    constexpr int KQ_ = (WW + 3) / 4, W4_ = 4 * KQ_, WS_ = (W4_ / 4) % 2 ? W4_ : W4_ + 4;                       \
    hipLaunchKernelGGL((xproj_bwd_mm_kernel<WW>), mgrid, dim3(256), (size_t)64 * WS_ * 4, mst, p, cbw);        \
  } while (0)
    switch (width) {
      case 44: FV_XM(44); break;
      case 56: FV_XM(56); break;
      case 80: FV_XM(80); break;
      case 96: FV_XM(96); break;
      case 112: FV_XM(112); break;
      case 34: FV_XM(34); break;
      case 36: FV_XM(36); break;
      case 38: FV_XM(38); break;
      case 64: FV_XM(64); break;
      default:
        fv_set_error("mixer_xproj_bwd: x_dbl width %d is not built", width);
        return FV_ERR_UNSUPPORTED;
    }
#undef FV_XM
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  const int bs = d_inner >= 256 ? 128 : 64;
  static const int rb = fv_tune("FASTVIM_XPROJ_RB", 8);   // tuning hook (rows of loads in flight; 8: 25.7 vs 26.8 us)
  dim3 grid(fv_cdiv(d_inner, bs), fv_mixer_xproj_bwd_slices(M), 2), block(bs);
  hipStream_t st = (hipStream_t)stream;
#define FV_XPK(TT, WW, RR)                                                                                     \
  do {                                                                                                         \
    if (dW_partials) hipLaunchKernelGGL((xproj_bwd_kernel<TT, WW, RR, true>), grid, block, (size_t)xproj_rows() * WW * 4, st, p);  \
    else hipLaunchKernelGGL((xproj_bwd_kernel<TT, WW, RR, false>), grid, block, (size_t)xproj_rows() * WW * 4, st, p);             \
  } while (0)
#define FV_XP(TT, WW)                                                                                          \
  do {                                                                                                         \
    if (rb == 8) FV_XPK(TT, WW, 8);                                                                            \
    else if (rb == 16) FV_XPK(TT, WW, 16);                                                                     \
    else FV_XPK(TT, WW, 4);                                                                                    \
  } while (0)
#define FV_XPD(WW) do { if (dtype == FV_F32) FV_XP(float, WW); else FV_XP(bf16_t, WW); } while (0)
  switch (width) {   // dt_rank + 2 * d_state for d_model = 192 / 384 / 768 / 1024 / 1280 (d_state 16) and small test models
    case 44: FV_XPD(44); break;
    case 56: FV_XPD(56); break;
    case 80: FV_XPD(80); break;
    case 96: FV_XPD(96); break;
    case 112: FV_XPD(112); break;
    case 34: FV_XPD(34); break;
    case 36: FV_XPD(36); break;
    case 38: FV_XPD(38); break;
    case 64: FV_XPD(64); break;
    default:
      fv_set_error("mixer_xproj_bwd: x_dbl width %d is not built", width);
      return FV_ERR_UNSUPPORTED;
  }
#undef FV_XPD
#undef FV_XP
#undef FV_XPK
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_mixer_xproj_fwd(const void* xc, const void* x_proj_w2, void* x_dbl, int M, int d_inner, int width,
                                  fv_stream_t stream) {
  FV_CHECK(xc && x_proj_w2 && x_dbl, "mixer_xproj_fwd: null pointer");
  FV_CHECK(M > 0 && width > 0 && width <= 112 && d_inner > 0 && d_inner % 32 == 0,
           "mixer_xproj_fwd: needs d_inner %% 32 == 0 and width <= 112 (got %d, %d)", d_inner, width);
  FV_CHECK(((uintptr_t)xc & 15) == 0 && ((uintptr_t)x_proj_w2 & 15) == 0, "mixer_xproj_fwd: operands must be 16-byte aligned");
  const bool wide = d_inner >= 768;           // more K per row block: eight waves
  const dim3 grid(fv_cdiv(M, 16), 2), block(wide ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
  const int nbk = fv_cdiv(width, 16);
#define FV_XF(NN)                                                                                                    \
  do {                                                                                                               \
    if (wide) hipLaunchKernelGGL((xproj_fwd_kernel<NN, 8>), grid, block, 0, st, (const bf16_t*)xc,                   \
                                 (const bf16_t*)x_proj_w2, (bf16_t*)x_dbl, M, d_inner, width);                       \
    else hipLaunchKernelGGL((xproj_fwd_kernel<NN, 4>), grid, block, 0, st, (const bf16_t*)xc,                        \
                            (const bf16_t*)x_proj_w2, (bf16_t*)x_dbl, M, d_inner, width);                            \
  } while (0)
  switch (nbk) {
    case 1: FV_XF(1); break;
    case 2: FV_XF(2); break;
    case 3: FV_XF(3); break;
    case 4: FV_XF(4); break;
    case 5: FV_XF(5); break;
    case 6: FV_XF(6); break;
    default: FV_XF(7); break;
  }
#undef FV_XF
  FV_LAUNCH_CHECK();
  return FV_OK;
}
