// hipcc-flags: -fno-slp-vectorize -fgpu-flush-denormals-to-zero
// Whole-row conv + pool (+ skip) forward for short pooling rows (cols == 14 or 16, tokens_per_patch == 1):
// both depthwise convs + SiLU, the pooling over the row and skip = D*conv_f + D_b*conv_b
// (mamba_simple_faster.py:272-305, 356-358, 412-416).  Written for instruction count like
// convpool_bwd_row.hip: explicit 2-wide packed math, compile-time token positions, buffer addressing,
// every load of the row (cols + 6 packed tokens) in flight before the first SiLU.
#include <stdlib.h>

#include "mixer_common.h"
#include "packed.h"

namespace {

using fvi::FwdParams;

template <typename T, int NT, int NP, bool PMAX>
__global__ __launch_bounds__(NP == 1 ? 1024 : 512) void conv_pool_fwd_row_kernel(FwdParams p) {
  typedef PairVec<T, NP> P;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x, b = blockIdx.y;
  const int c0 = ((blockIdx.z * (blockDim.x >> 6) + wv) * 64 + lane) * 2 * NP;           // first channel of this lane
  const Geo g = p.geo;
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_s = p.d_in * (int)sizeof(T);
  const int voff = c0 * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t bx = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in, (size_t)g.L * tok_x);
  const __amdgpu_buffer_rsrc_t bs = fv_make_buf((T*)p.skip + (size_t)b * g.L * p.d_in, p.skip ? (size_t)g.L * tok_s : 0);
  const int m_row = i * g.s_i;
  const bool up = i > 0, down = i + 1 < g.rows;
  const int s_up = up ? -g.s_i : 0, s_dn = down ? g.s_i : 0;
  P xr[NT + 6];
#pragma unroll
  for (int k = 0; k < NT + 6; ++k) {
    const int di = k < 3 ? -1 : (k >= NT + 3 ? 1 : 0);
    const int j = k - 3 - di * NT;
    xr[k].load(bx, voff, (m_row + (di < 0 ? s_up : di > 0 ? s_dn : 0) + j * g.s_j) * tok_x);
  }
  f2 wf[NP][CW], wb[NP][CW], bf[NP], bb[NP], Df[NP], Db[NP], accf[NP], accb[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    load_taps2(p.wf, c0 + 2 * q, wf[q]);
    load_taps2(p.wb, c0 + 2 * q, wb[q]);
    bf[q] = load_f2(p.bf, c0 + 2 * q);
    bb[q] = load_f2(p.bb, c0 + 2 * q);
    Df[q] = load_f2(p.skip ? p.Df : nullptr, c0 + 2 * q);
    Db[q] = load_f2(p.skip ? p.Db : nullptr, c0 + 2 * q);
    accf[q] = accb[q] = splat(PMAX ? -INFINITY : 0.f);
  }
  const float m_up = up ? 1.f : 0.f, m_dn = down ? 1.f : 0.f;
  f2 x[NT + 6][NP];                      // index q + 3; live ranges are 7 steps (full unroll)
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int q = 0; q < NP; ++q) x[k][q] = k < 3 ? xr[k].get(q) * m_up : xr[k].get(q);
#pragma unroll
  for (int jj = 0; jj < NT; ++jj) {
    f2 sk[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      x[jj + 6][q] = xr[jj + 6].get(q);
      if (jj + 3 >= NT) x[jj + 6][q] *= m_dn;
      f2 pf = bf[q], pb = bb[q];
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        pf = fma2(wf[q][k], x[jj + k][q], pf);            // x[s-3+k]
        pb = fma2(wb[q][k], x[jj + 6 - k][q], pb);        // x[s+3-k]
      }
      const f2 xf = silu2(pf), xb = silu2(pb);
      sk[q] = fma2(Df[q], xf, Db[q] * xb);
      if (PMAX) {
        accf[q] = __builtin_elementwise_max(accf[q], xf);
        accb[q] = __builtin_elementwise_max(accb[q], xb);
      } else {
        accf[q] += xf;
        accb[q] += xb;
      }
    }
    if (p.skip) P::store(bs, voff, (m_row + jj * g.s_j) * tok_s, sk);
  }
  T* xc = (T*)p.xc;
  const size_t dstride = (size_t)p.B * g.rows * p.d_in;
  const size_t o = ((size_t)b * g.rows + i) * p.d_in + c0;
  float of[2 * NP], ob[2 * NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const f2 a = accf[q] * p.pool_scale, c = accb[q] * p.pool_scale;
    of[2 * q] = a.x; of[2 * q + 1] = a.y;
    ob[2 * q] = c.x; ob[2 * q + 1] = c.y;
  }
  VecIO<T, 2 * NP>::store(xc + o, of);
  VecIO<T, 2 * NP>::store(xc + dstride + o, ob);
}

// Cell-walking kernel for rows that are too long to hold in registers.  A pooling row is walked in cells of TPP = 8
// tokens whose positions inside the cell are compile-time (register windows, affine cell addresses -- no division, no
// LDS), two cells of packed loads always in flight ahead of the arithmetic.
//   CHAN  (channel-wise tokenization, tokens_per_patch == TPP, Channel-First): a cell is one patch, token c of every
//         cell pools into slot c (TPP slot accumulators in registers);
//   !CHAN (tokens_per_patch == 1, cols a multiple of TPP -- the 512 / 1024 / 2048 px grids): a cell is TPP consecutive
//         positions of the row, everything pools into one slot.
// Channels beyond 1024 are split over blockIdx.z (a lane owns a channel pair, a block at most 8 waves).
template <typename T, int TPP, bool PMAX, bool CHAN>
__global__ __launch_bounds__(512) void conv_pool_fwd_chan_kernel(FwdParams p) {
  typedef PairVec<T, 1> P;
  static_assert(TPP >= 3, "the conv halo (3 tokens) must fit in one neighbouring cell");
  constexpr int NS = CHAN ? TPP : 1;                     // pooling slots per row
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x, b = blockIdx.y;
  const int c0 = ((blockIdx.z * (blockDim.x >> 6) + wv) * 64 + lane) * 2;
  const Geo g = p.geo;
  const int ncell = CHAN ? g.pcols : g.cols / TPP;
  const int ts = CHAN ? 1 : g.s_j;                       // memory tokens between consecutive tokens of a cell
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_s = p.d_in * (int)sizeof(T);
  const int voff = c0 * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t bx = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in, (size_t)g.L * tok_x);
  const __amdgpu_buffer_rsrc_t bs = fv_make_buf((T*)p.skip + (size_t)b * g.L * p.d_in, p.skip ? (size_t)g.L * tok_s : 0);
  const bool up = i > 0, down = i + 1 < g.rows;
  // first memory token of cell jj of this row; jj = -1 / ncell are the neighbouring rows' last / first cell
  // (a missing neighbour row reads this row's own cell instead -- always mapped -- and is masked to zero)
  auto cell = [&](int jj) {
    int ri = i, cj = jj;
    if (jj < 0) { ri = up ? i - 1 : i; cj = ncell - 1; }
    else if (jj >= ncell) { ri = down ? i + 1 : i; cj = 0; }
    return CHAN ? (ri * g.s_i + cj * g.s_j) * TPP : ri * g.s_i + cj * TPP * g.s_j;
  };
  f2 wf[CW], wb[CW];
  load_taps2(p.wf, c0, wf);
  load_taps2(p.wb, c0, wb);
  const f2 bf = load_f2(p.bf, c0), bb = load_f2(p.bb, c0);
  const f2 Df = load_f2(p.skip ? p.Df : nullptr, c0), Db = load_f2(p.skip ? p.Db : nullptr, c0);
  f2 accf[NS], accb[NS];
#pragma unroll
  for (int c = 0; c < NS; ++c) accf[c] = accb[c] = splat(PMAX ? -INFINITY : 0.f);
  P r1[TPP], r2[TPP];                    // packed cells j+1 and j+2
  f2 prev[3], cur[TPP];
  {
    P rp[3], rc[TPP];
    const int mp = cell(-1), mc = cell(0), m1 = cell(1), m2 = cell(2);
#pragma unroll
    for (int k = 0; k < 3; ++k) rp[k].load(bx, voff, (mp + (TPP - 3 + k) * ts) * tok_x);
#pragma unroll
    for (int c = 0; c < TPP; ++c) rc[c].load(bx, voff, (mc + c * ts) * tok_x);
#pragma unroll
    for (int c = 0; c < TPP; ++c) r1[c].load(bx, voff, (m1 + c * ts) * tok_x);
#pragma unroll
    for (int c = 0; c < TPP; ++c) r2[c].load(bx, voff, (m2 + c * ts) * tok_x);
    const float m_up = up ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) prev[k] = rp[k].get(0) * m_up;
#pragma unroll
    for (int c = 0; c < TPP; ++c) cur[c] = rc[c].get(0);
  }
  // One body per cell; rn = the packed cell j + 1, consumed at the end of the body and refilled there with cell j + 3.
  // The two packed buffers alternate between bodies (loop unrolled by two): written as `r1 = r2; r2 = load` the rotation
  // is a parallel copy at the bottom of the loop, the loads land in temporaries and the copies out of them wait for
  // every load of the iteration that issued it (round 6, ISA of the backward twin: the cell "two ahead" was waited for
  // at once).
  auto body = [&](int j, P (&rn)[TPP]) {
    // window of cell j: prev[0..2] | cur[0..TPP) | next[0..2]
    const float m_nx = (j + 1 < ncell || down) ? 1.f : 0.f;
    f2 xw[TPP + 6];
#pragma unroll
    for (int k = 0; k < 3; ++k) xw[k] = prev[k];
#pragma unroll
    for (int c = 0; c < TPP; ++c) xw[3 + c] = cur[c];
#pragma unroll
    for (int k = 0; k < 3; ++k) xw[3 + TPP + k] = rn[k].get(0) * m_nx;
    const int mc = cell(j);
#pragma unroll
    for (int c = 0; c < TPP; ++c) {
      f2 pf = bf, pb = bb;
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        pf = fma2(wf[k], xw[c + k], pf);             // x[s-3+k]
        pb = fma2(wb[k], xw[c + 6 - k], pb);         // x[s+3-k]
      }
      const f2 xf = silu2(pf), xb = silu2(pb);
      if (p.skip) {
        const f2 sk[1] = {fma2(Df, xf, Db * xb)};
        P::store(bs, voff, (mc + c * ts) * tok_s, sk);
      }
      constexpr int sl = CHAN ? 1 : 0;               // slot of token c: c (channel path) or 0
      if (PMAX) {
        accf[c * sl] = __builtin_elementwise_max(accf[c * sl], xf);
        accb[c * sl] = __builtin_elementwise_max(accb[c * sl], xb);
      } else {
        accf[c * sl] += xf;
        accb[c * sl] += xb;
      }
    }
    // rotate: cell j+1 becomes current, j+2 waits, j+3 is fetched
#pragma unroll
    for (int k = 0; k < 3; ++k) prev[k] = cur[TPP - 3 + k];
#pragma unroll
    for (int c = 0; c < TPP; ++c) cur[c] = rn[c].get(0) * m_nx;
    const int m3 = cell(j + 3 > ncell ? ncell : j + 3);
#pragma unroll
    for (int c = 0; c < TPP; ++c) rn[c].load(bx, voff, (m3 + c * ts) * tok_x);
  };
  for (int j = 0; j < ncell; j += 2) {
    body(j, r1);
    if (j + 1 < ncell) body(j + 1, r2);
  }
  T* xc = (T*)p.xc;
  const size_t dstride = (size_t)p.B * g.rows * NS * p.d_in;
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    const size_t o = (((size_t)b * g.rows + i) * NS + c) * p.d_in + c0;
    const f2 a = accf[c] * p.pool_scale, d = accb[c] * p.pool_scale;
    const float of[2] = {a.x, a.y}, ob[2] = {d.x, d.y};
    VecIO<T, 2>::store(xc + o, of);
    VecIO<T, 2>::store(xc + dstride + o, ob);
  }
}

// channel groups: the smallest split of d_in / 128 waves into blocks of at most 2 waves.  The waves of a block share
// nothing (each owns 128 channels of the row), so small blocks only help the dispatcher fill the CUs: FastChannelVim-S
// (6 waves of channels) 101.1 us with one 6-wave block per row, 86.9 with three 2-wave blocks; FastVim-B at 2048 px
// (12 waves) 239.6 -> 195.7 us (profiles/r05_ab_chan_block_shapes.log)
inline int chan_groups(int d_in) {
  const int nw = d_in / 128;
  int gq = (nw + 1) / 2;
  while (nw % gq) ++gq;
  return gq;
}

template <typename T, int TPP, bool CHAN>
int launch_chan(const FwdParams& p, int pool_max, hipStream_t st) {
  static const int t_gq = fv_tune("FASTVIM_FWD_CHAN_GROUPS", 0);   // tuning hook
  const int nw = p.d_in / 128;
  const int gq = (t_gq > 0 && nw % t_gq == 0) ? t_gq : chan_groups(p.d_in), nch = nw / gq;
  dim3 grid(p.geo.rows, p.B, gq), block(64 * nch);
  if (pool_max) hipLaunchKernelGGL((conv_pool_fwd_chan_kernel<T, TPP, true, CHAN>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((conv_pool_fwd_chan_kernel<T, TPP, false, CHAN>), grid, block, 0, st, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T, int NT, int NP>
int launch_row(const FwdParams& p, int pool_max, hipStream_t st) {
  static const int t_gq = fv_tune("FASTVIM_FWD_ROW_GROUPS", 0);   // tuning hook
  // blocks of at most 4 waves (channel groups over blockIdx.z), like the long-row kernel: FastVim-B 46.4 -> 43.2 us,
  // FastVim-T (3 waves) unchanged
  const int nw = p.d_in / (128 * NP);
  int gq = (nw + 3) / 4;
  while (nw % gq) ++gq;
  if (t_gq > 0 && nw % t_gq == 0) gq = t_gq;
  const int nch = nw / gq;
  dim3 grid(p.geo.rows, p.B, gq), block(64 * nch);
  if (pool_max) hipLaunchKernelGGL((conv_pool_fwd_row_kernel<T, NT, NP, true>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((conv_pool_fwd_row_kernel<T, NT, NP, false>), grid, block, 0, st, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T, int NT>
int pick_np(const FwdParams& p, int pool_max, hipStream_t st) {
  static const int force = fv_tune("FASTVIM_FWD_NP", 0);   // tuning hook
  // one channel pair per lane measured fastest (12.8 vs 16.5 us with three pairs on FastVim-T): more, shorter waves
  if ((force == 0 || force == 1) && p.d_in % 128 == 0 && p.d_in <= 16 * 128) return launch_row<T, NT, 1>(p, pool_max, st);
  // (three pairs per lane -- d_inner 2304 ... 3072 only -- spilled 26-62 registers at 256 and served no model: those widths
  //  take the generic kernel)
  if ((force == 0 || force == 2) && p.d_in % 256 == 0 && p.d_in <= 8 * 256) return launch_row<T, NT, 2>(p, pool_max, st);
  return FV_ERR_UNSUPPORTED;
}

}  // namespace

int fvi::conv_pool_fwd_row(const FwdParams& p, int pool_max, int dtype, hipStream_t st) {
  const bool fits = p.d_in % 128 == 0 && (size_t)p.geo.L * 2 * p.d_in * 4 <= 0xfffff000ull;   // one batch element per descriptor
  static const bool chan = (fv_tune("FASTVIM_FWD_CHAN", 1) != 0);   // tuning hook
  if (chan && fits && !pool_max && p.geo.tpp == 8 && p.geo.pcols >= 2 && p.d_in <= 8 * 128)
    return dtype == FV_F32 ? launch_chan<float, 8, true>(p, pool_max, st) : launch_chan<bf16_t, 8, true>(p, pool_max, st);
  // long rows of the dense path (cols = 32 / 64 / 128: the 512 / 1024 / 2048 px grids)
  if (chan && fits && !pool_max && p.geo.tpp == 1 && p.geo.cols % 8 == 0 && p.geo.cols >= 24)
    return dtype == FV_F32 ? launch_chan<float, 8, false>(p, pool_max, st) : launch_chan<bf16_t, 8, false>(p, pool_max, st);
  if (p.geo.tpp != 1 || (p.geo.cols != 14 && p.geo.cols != 16)) return FV_ERR_UNSUPPORTED;
  if (!fits) return FV_ERR_UNSUPPORTED;
  if (dtype == FV_F32) return p.geo.cols == 14 ? pick_np<float, 14>(p, pool_max, st) : pick_np<float, 16>(p, pool_max, st);
  return p.geo.cols == 14 ? pick_np<bf16_t, 14>(p, pool_max, st) : pick_np<bf16_t, 16>(p, pool_max, st);
}
