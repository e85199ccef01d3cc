// hipcc-flags: -fno-slp-vectorize -fgpu-flush-denormals-to-zero
// Whole-row conv + pool backward for short pooling rows (cols == 14 or 16, tokens_per_patch == 1,
// d_inner a multiple of 128): the adjoint of the D-skip, mean-pool, SiLU and both depthwise convs
// (mamba_simple_faster.py:270-305, 356-358, 412-416; FastVim_MambaInnerFnNoOutProj_withoutZ.backward,
// selective_scan_interface.py:607-776).  Same persistent grid and per-block partial layout as the generic
// kernel in mixer_bwd.hip.
//
// These launches are instruction-issue bound (two sigmoids = four quarter-rate transcendentals per element
// and ~50 FMAs), so the kernel is written for instruction count:
//   * a lane owns a channel PAIR and all per-channel arithmetic is on 2-wide vectors -> v_pk_fma/mul/add_f32
//     (the SLP vectoriser is off for this file: left to itself it pairs values across the unrolled token
//     steps instead and drowns in splat copies);
//   * all positions are compile-time: token offsets are affine (no integer division), the 4-deep windows are
//     register renaming;
//   * token accesses are buffer instructions: descriptor + uniform byte offset in SGPRs and one shared
//     32-bit lane offset, instead of a 64-bit VGPR address pair per access;
//   * every load of the row (2 x (cols + 6) tokens, packed) is issued before the first sigmoid; halo
//     positions of a missing neighbour row read this row (always mapped) and are zeroed by scalar selects.
#include "mixer_common.h"
#include "packed.h"

namespace {

using fvi::BwdParams;

// One partial row per block: [d w (d_in*4) | d w_b (d_in*4) | d b | d b_b | dD | dD_b]; the row groups of a block are
// summed through LDS in a fixed order.  A block owns the channels [c_lo, c_lo + nc) (all of them unless the channels
// are split over blockIdx.y) and writes only their entries of the row.
__device__ __forceinline__ void flush_partials(const BwdParams& p, float* smem, int c0, int c_lo, int nc, int rg, int RG,
                                               const f2 (&a_wf)[CW], const f2 (&a_wb)[CW], f2 a_bf, f2 a_bb, f2 a_Df, f2 a_Db) {
  const int D = p.d_in;
  for (int r = 0; r < RG; ++r) {
    __syncthreads();
    if (r == rg) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int c = c0 - c_lo + v;          // local channel
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          smem[c * 4 + k] = (r == 0 ? 0.f : smem[c * 4 + k]) + a_wf[k][v];
          smem[4 * nc + c * 4 + k] = (r == 0 ? 0.f : smem[4 * nc + c * 4 + k]) + a_wb[k][v];
        }
        smem[8 * nc + c] = (r == 0 ? 0.f : smem[8 * nc + c]) + a_bf[v];
        smem[9 * nc + c] = (r == 0 ? 0.f : smem[9 * nc + c]) + a_bb[v];
        smem[10 * nc + c] = (r == 0 ? 0.f : smem[10 * nc + c]) + a_Df[v];
        smem[11 * nc + c] = (r == 0 ? 0.f : smem[11 * nc + c]) + a_Db[v];
      }
    }
  }
  __syncthreads();
  float* dst = p.part + (size_t)blockIdx.x * 12 * D;
  for (int e = threadIdx.x; e < 12 * nc; e += blockDim.x) {
    int gi;
    if (e < 4 * nc) gi = c_lo * 4 + e;
    else if (e < 8 * nc) gi = 4 * D + c_lo * 4 + (e - 4 * nc);
    else {
      const int q = (e - 8 * nc) / nc, r = (e - 8 * nc) - q * nc;
      gi = (8 + q) * D + c_lo + r;
    }
    dst[gi] = smem[e];
  }
}

// Where a row lives: descriptors of its batch element and the token offsets of the row and of its two neighbours.
// A row past the end gets zero-length descriptors: its (prefetch) loads return zeros and touch no memory.
struct RowSrc {
  __amdgpu_buffer_rsrc_t bx, bd;
  int pooled_off;            // byte offset of row i's forward pooled gradient in dxc
  int m_row, s_up, s_dn;
  bool up, down;
  int b;
};

// X2: the pooled gradient is dxc + dxc2 (dxc2 in the storage dtype: the other channel chunk's x_proj term, written by
// fv_mixer_scan_bwd_xproj); the second addend travels packed, one row ahead like the first, and is added at first use
template <typename T, int NT, bool X2 = false>
__global__ __launch_bounds__(sizeof(T) == 2 && NT <= 14 ? 768 : 512) void conv_pool_bwd_row_kernel(BwdParams p, int nch, int RG) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 12 * d_in accumulator
  typedef PairVec<T, 1> P;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wv / nch, cw = wv - rg * nch;
  const int c_lo = blockIdx.y * nch * 128;             // channels beyond 1024 are split over blockIdx.y
  const int c0 = c_lo + (cw * 64 + lane) * 2;          // first channel of this lane's pair
  const Geo g = p.geo;
  f2 a_wf[CW], a_wb[CW], a_bf = splat(0.f), a_bb = splat(0.f), a_Df = splat(0.f), a_Db = splat(0.f);
#pragma unroll
  for (int k = 0; k < CW; ++k) a_wf[k] = a_wb[k] = splat(0.f);
  const int nrows = p.B * g.rows;
  const size_t dstride = (size_t)p.B * g.rows * p.d_in;
  const int nit = (nrows + gridDim.x * RG - 1) / (gridDim.x * RG);
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_d = p.d_in * (int)sizeof(T);   // bytes per token
  const int voff = c0 * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t bp = fv_make_buf(p.dxc, 2 * dstride * 4);    // both directions' pooled gradients
  const __amdgpu_buffer_rsrc_t bp2 = fv_make_buf(X2 ? p.dxc2 : p.dxc, X2 ? 2 * dstride * sizeof(T) : 0);
  struct Pooled {          // a pooled gradient as loaded: fp32 pair + (X2) the packed second addend
    f2 a;
    P b;
    __device__ __forceinline__ f2 sum() const { if constexpr (X2) return a + b.get(0); else return a; }
  };
  auto locate = [&](int row) {
    const bool ok = row < nrows;
    const int r = ok ? row : 0;
    const int b = r / g.rows, i = r - b * g.rows;
    RowSrc s;
    s.b = b;
    s.bx = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in, ok ? (size_t)g.L * tok_x : 0);
    s.bd = fv_make_buf((const T*)p.dob_in + (size_t)b * g.L * p.d_in, ok ? (size_t)g.L * tok_d : 0);
    s.up = i > 0;
    s.down = i + 1 < g.rows;
    s.m_row = i * g.s_i;
    s.s_up = s.up ? -g.s_i : 0;            // a missing neighbour row reads this row (always mapped) and is masked
    s.s_dn = s.down ? g.s_i : 0;
    s.pooled_off = (b * g.rows + i) * p.d_in * 4;
    return s;
  };
  // packed tokens, positions q = -3 .. NT+2 of a row (index q + 3).  The row being worked on was loaded while the
  // previous one was: every register is refilled with the NEXT row's token as soon as this row has unpacked it, so a
  // wave always has a whole row of loads in flight under its arithmetic.
  P xr[NT + 6], dr[NT + 6];
  auto fetch = [&](const RowSrc& s, int k) {
    const int di = k < 3 ? -1 : (k >= NT + 3 ? 1 : 0);
    const int j = k - 3 - di * NT;
    const int m = s.m_row + (di < 0 ? s.s_up : di > 0 ? s.s_dn : 0) + j * g.s_j;
    xr[k].load(s.bx, voff, m * tok_x);
    dr[k].load(s.bd, voff, m * tok_d);
  };
  // pooled gradient of row i - 1 + r around row s (unscaled: the multiply by pool_scale, or by 0 for a missing row,
  // waits for the load and is placed at the first use)
  auto pooled = [&](const RowSrc& s, int r, bool backward) {
    const bool ok = r == 1 || (r == 0 ? s.up : s.down);
    uint32_t w[2];
    const int eoff = s.pooled_off / 4 + (ok ? (r - 1) * p.d_in : 0) + (backward ? (int)dstride : 0);     // element offset
    fv_buf_load_words<2>(bp, c0 * 4, eoff * 4, w);
    Pooled o;
    o.a.x = __uint_as_float(w[0]);
    o.a.y = __uint_as_float(w[1]);
    if constexpr (X2) o.b.load(bp2, voff, eoff * (int)sizeof(T));
    return o;
  };
  auto row_of = [&](int it) { return (it * gridDim.x + blockIdx.x) * RG + rg; };
  RowSrc cur = locate(row_of(0));
  // The per-channel parameters are requested FIRST and the first row right behind them, with no wait in between: loads
  // return in order, so the arithmetic of step -3 starts when the parameters and the first four tokens are there and
  // the rest of the row streams in under it.  (Requested last, their wait was a wait for the whole row -- and every wave
  // of the grid asks for its first row at the same moment: 35 MB, 7 us in which nothing was computed.)
  // (an absent bias / D is read from the taps and multiplied by zero: no branch around the load, no wait inside it)
  f2 wf[CW], wb[CW];
  load_taps2(p.wf, c0, wf);
  load_taps2(p.wb, c0, wb);
  const f2 bf_raw = load_f2(p.bf ? p.bf : p.wf, c0), bb_raw = load_f2(p.bb ? p.bb : p.wf, c0);
  const f2 Df_raw = load_f2(p.Df ? p.Df : p.wf, c0), Db_raw = load_f2(p.Db ? p.Db : p.wf, c0);
  __builtin_amdgcn_sched_barrier(0);
  // Pooled gradients: the forward conv needs rows i (positions of this row) and i+1 (the three halo positions
  // behind it), the backward conv rows i-1 and i.  All are loaded unscaled, one row ahead, in the order of their
  // first use -- here exactly as in the loop, so that the waits the compiler places at the top of the loop body (one
  // count for both ways in) find them oldest: row i's pair at the top of the previous iteration, row i-1's after its
  // last use (step -1), row i+1's at the end.
  Pooled cf_mid_raw = pooled(cur, 1, false), cb_mid_raw = pooled(cur, 1, true);
  Pooled cb_up_raw = pooled(cur, 0, true);
  __builtin_amdgcn_sched_barrier(0);      // (the scheduler would issue these last)
#pragma unroll
  for (int k = 0; k < NT + 6; ++k) fetch(cur, k);
  __builtin_amdgcn_sched_barrier(0);
  Pooled cf_dn_raw = pooled(cur, 2, false);
  __builtin_amdgcn_sched_barrier(0);
  const f2 bf = bf_raw * (p.bf ? 1.f : 0.f), bb = bb_raw * (p.bb ? 1.f : 0.f);
  const f2 Dfh = Df_raw * (p.Df ? 0.5f : 0.f), Dbh = Db_raw * (p.Db ? 0.5f : 0.f);
  for (int it = 0; it < nit; ++it) {
    if (row_of(it) >= nrows) break;          // uniform per wave; no block-level sync inside
    const RowSrc nxt = locate(row_of(it + 1));
    const __amdgpu_buffer_rsrc_t bo = fv_make_buf((T*)p.dxz + (size_t)cur.b * g.L * 2 * p.d_in, (size_t)g.L * tok_x);
    const bool up = cur.up, down = cur.down;
    const int m_row = cur.m_row;
    const f2 cf_mid = cf_mid_raw.sum() * p.pool_scale, cb_mid = cb_mid_raw.sum() * p.pool_scale;
    cf_mid_raw = pooled(nxt, 1, false);
    cb_mid_raw = pooled(nxt, 1, true);
    const f2 cb_up = cb_up_raw.sum() * (cur.up ? p.pool_scale : 0.f);
    f2 cf_dn = splat(0.f);
    f2 x[NT + 6], dov[NT + 6], dpf[NT + 6], dpb[NT + 6];   // index q + 3; live ranges are 4 steps (full unroll)
    const float m_up = up ? 1.f : 0.f, m_dn = down ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      x[k] = xr[k].get(0) * m_up;
      dov[k] = dr[k].get(0);
      dpf[k] = dpb[k] = splat(0.f);
      fetch(nxt, k);
    }
#pragma unroll
    for (int n = -3; n < NT; ++n) {
      // step n: pre_f of position n+3 and pre_b of position n, both from x[n .. n+3]
      const int q3 = n + 3, k0 = n + 3, k3 = n + 6;             // array indices of positions n and n+3
      x[k3] = xr[k3].get(0);
      if (q3 >= NT) x[k3] *= m_dn;
      dov[k3] = dr[k3].get(0);
      fetch(nxt, k3);
      f2 pf = bf, pb = bb;
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        pf = fma2(wf[k], x[k0 + k], pf);             // pre_f[n+3] = b + sum_k w[k] x[n+k]
        pb = fma2(wb[k], x[k3 - k], pb);             // pre_b[n]   = b + sum_k w[k] x[n+3-k]
      }
      const f2 sgf = sigmoid2(pf), sgb = sigmoid2(pb);
      const f2 dsf = sgf * fma2(pf, 1.f - sgf, splat(1.f)), dsb = sgb * fma2(pb, 1.f - sgb, splat(1.f));
      // position n+3 / n missing (outside the sequence): its gradient is zero
      const float e3 = q3 < NT ? 1.f : m_dn, e0 = n >= 0 ? 1.f : m_up;
      if (q3 == NT) cf_dn = cf_dn_raw.sum() * (cur.down ? p.pool_scale : 0.f);
      const f2 nf = fma2(Dfh, dov[k3], q3 >= NT ? cf_dn : cf_mid) * dsf * e3;
      const f2 nb = fma2(Dbh, dov[k0], n < 0 ? cb_up : cb_mid) * dsb * e0;
      if (n == -1) cb_up_raw = pooled(nxt, 0, true);
      dpf[k3] = nf;
      dpb[k0] = nb;
      if (q3 < NT) {        // position n+3 belongs to this row: its parameter gradients are accumulated here
#pragma unroll
        for (int k = 0; k < CW; ++k) a_wf[k] = fma2(nf, x[k0 + k], a_wf[k]);
        a_bf += nf;
        a_Df = fma2(dov[k3], pf * sgf, a_Df);            // x 0.5 once, at the flush (exact)
      }
      if (n >= 0) {
#pragma unroll
        for (int k = 0; k < CW; ++k) a_wb[k] = fma2(nb, x[k3 - k], a_wb[k]);
        a_bb += nb;
        a_Db = fma2(dov[k0], pb * sgb, a_Db);
        // dx[n] = sum_k wf[k] dpre_f[n+3-k] + wb[k] dpre_b[n-3+k]
        f2 dx = splat(0.f);
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          dx = fma2(wf[k], dpf[k3 - k], dx);
          dx = fma2(wb[k], dpb[k0 - 3 + k], dx);
        }
        { const f2 dxa[1] = {dx}; P::store(bo, voff, (m_row + n * g.s_j) * tok_x, dxa); }
      }
      // the prefetch stays in the step that freed its registers: left alone, the scheduler sinks every load of the
      // iteration to its end (shorter live ranges) and the next iteration starts by waiting for all of them
      __builtin_amdgcn_sched_barrier(0);
    }
    cf_dn_raw = pooled(nxt, 2, false);
    cur = nxt;
  }
  flush_partials(p, smem, c0, c_lo, nch * 128, rg, RG, a_wf, a_wb, a_bf, a_bb, a_Df * 0.5f, a_Db * 0.5f);
}

// Long rows, walked cell by cell like conv_pool_fwd_chan_kernel: CHAN = channel-wise tokenization (tokens_per_patch ==
// TPP, Channel-First; mamba_simple_channel_faster.py:242-256, 333-340), !CHAN = the dense path with cols a multiple
// of TPP (one pooling slot per row; the 512 / 1024 / 2048 px grids).  Channels beyond 1024 are split over blockIdx.y.  Body j runs the TPP steps n = TPP*j - 3 + c, so the
// pooling slots of both positions a step touches (n + 3 -> slot c; n -> slot c - 3 of this cell or TPP - 3 + c of the
// previous one) are compile-time and the slot gradients stay in registers; one extra 3-step body closes the row.
template <typename T, int TPP, bool CHAN>
__global__ __launch_bounds__(512) void conv_pool_bwd_chan_kernel(BwdParams p, int nch, int RG) {
  constexpr int NS = CHAN ? TPP : 1;          // pooling slots per row
  constexpr int NH = CHAN ? 3 : 1;            // slots of a neighbouring row that the 3-token halo touches
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 12 * d_in accumulator
  typedef PairVec<T, 1> P;
  static_assert(TPP >= 6, "slot bookkeeping assumes the three carried tokens and the three halo tokens do not overlap");
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wv / nch, cw = wv - rg * nch;
  const int c_lo = blockIdx.y * nch * 128;
  const int c0 = c_lo + (cw * 64 + lane) * 2;
  const Geo g = p.geo;
  const int pcols = CHAN ? g.pcols : g.cols / TPP;      // cells per row
  const int ts = CHAN ? 1 : g.s_j;                      // memory tokens between consecutive tokens of a cell
  f2 wf[CW], wb[CW], bf, bb, Dfh, Dbh;
  load_taps2(p.wf, c0, wf);
  load_taps2(p.wb, c0, wb);
  bf = load_f2(p.bf, c0);
  bb = load_f2(p.bb, c0);
  Dfh = load_f2(p.Df, c0) * 0.5f;
  Dbh = load_f2(p.Db, c0) * 0.5f;
  f2 a_wf[CW], a_wb[CW], a_bf = splat(0.f), a_bb = splat(0.f), a_Df = splat(0.f), a_Db = splat(0.f);
#pragma unroll
  for (int k = 0; k < CW; ++k) a_wf[k] = a_wb[k] = splat(0.f);
  const int nrows = p.B * g.rows;
  const size_t dstride = (size_t)p.B * g.rows * NS * p.d_in;
  const int nit = (nrows + gridDim.x * RG - 1) / (gridDim.x * RG);
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_d = p.d_in * (int)sizeof(T);
  const int voff = c0 * (int)sizeof(T);
  for (int it = 0; it < nit; ++it) {
    const int row = (it * gridDim.x + blockIdx.x) * RG + rg;
    if (row < nrows) {          // uniform per wave; no block-level sync inside
      const int b = row / g.rows, i = row - b * g.rows;
      const __amdgpu_buffer_rsrc_t bx = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in, (size_t)g.L * tok_x);
      const __amdgpu_buffer_rsrc_t bd = fv_make_buf((const T*)p.dob_in + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_d);
      const __amdgpu_buffer_rsrc_t bo = fv_make_buf((T*)p.dxz + (size_t)b * g.L * 2 * p.d_in, (size_t)g.L * tok_x);
      const bool up = i > 0, down = i + 1 < g.rows;
      const float m_up = up ? 1.f : 0.f, m_dn = down ? 1.f : 0.f;
      // first memory token of cell jj; jj = -1 / pcols: the neighbouring rows' last / first cell (a missing neighbour
      // reads this row's own cell -- always mapped -- and is masked)
      auto cell = [&](int jj) {
        int ri = i, cj = jj;
        if (jj < 0) { ri = up ? i - 1 : i; cj = pcols - 1; }
        else if (jj >= pcols) { ri = down ? i + 1 : i; cj = 0; }
        return CHAN ? (ri * g.s_i + cj * g.s_j) * TPP : ri * g.s_i + cj * TPP * g.s_j;
      };
      // pooled gradients (x pool_scale): every slot of this row, the slots of row i+1 that its first three tokens
      // pool into (forward conv halo) and the slots of row i-1 that its last three tokens pool into (backward halo)
      // (CHAN: the six halo slot gradients are used in the first / last cell of a row only; they wait in a lane-private LDS
      //  slot instead of 12 registers that the allocator would otherwise spill to scratch at 256 VGPRs)
      f2 dcf[NS], dcb[NS], dcf_dn[NH], dcb_up[NH];
      f2* park = reinterpret_cast<f2*>(smem + 12 * nch * 128) + wv * ((2 * NH + 2 * NS) * 64) + lane;
      {
        const float* dq = p.dxc + ((size_t)b * g.rows + i) * NS * p.d_in + c0;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          dcf[c] = *reinterpret_cast<const f2*>(dq + (size_t)c * p.d_in) * p.pool_scale;
          dcb[c] = *reinterpret_cast<const f2*>(dq + dstride + (size_t)c * p.d_in) * p.pool_scale;
        }
        const float* dn = dq + (down ? (size_t)NS * p.d_in : 0);
        const float* du = dq + dstride - (up ? (size_t)NS * p.d_in : 0);
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          dcf_dn[k] = *reinterpret_cast<const f2*>(dn + (size_t)k * p.d_in) * (p.pool_scale * m_dn);
          dcb_up[k] = *reinterpret_cast<const f2*>(du + (size_t)(NS - NH + k) * p.d_in) * (p.pool_scale * m_up);
          if constexpr (CHAN) {
            park[k * 64] = dcf_dn[k];
            park[(NH + k) * 64] = dcb_up[k];
          }
        }
        if constexpr (CHAN) {
#pragma unroll
          for (int c = 0; c < NS; ++c) {
            park[(2 * NH + c) * 64] = dcf[c];
            park[(2 * NH + NS + c) * 64] = dcb[c];
          }
        }
      }
      // body arrays: X / DO / PF index = position - (TPP*j - 3); PB index = position - (TPP*j - 6)
      f2 X[TPP + 3], DO[TPP + 3], PF[TPP + 3], PB[TPP + 3];
      P xr1[TPP], xr2[TPP], dr1[TPP], dr2[TPP];      // packed cells j+1, j+2
      int m_prev = cell(-1), m_cur = cell(0);
      {
        P xa[3], da[3], xc[TPP], dc[TPP];
        const int m1 = cell(1), m2 = cell(2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          xa[k].load(bx, voff, (m_prev + (TPP - 3 + k) * ts) * tok_x);
          da[k].load(bd, voff, (m_prev + (TPP - 3 + k) * ts) * tok_d);
        }
#pragma unroll
        for (int c = 0; c < TPP; ++c) {
          xc[c].load(bx, voff, (m_cur + c * ts) * tok_x);
          dc[c].load(bd, voff, (m_cur + c * ts) * tok_d);
        }
#pragma unroll
        for (int c = 0; c < TPP; ++c) {
          xr1[c].load(bx, voff, (m1 + c * ts) * tok_x);
          dr1[c].load(bd, voff, (m1 + c * ts) * tok_d);
        }
#pragma unroll
        for (int c = 0; c < TPP; ++c) {
          xr2[c].load(bx, voff, (m2 + c * ts) * tok_x);
          dr2[c].load(bd, voff, (m2 + c * ts) * tok_d);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          X[k] = xa[k].get(0) * m_up;
          DO[k] = da[k].get(0);
          PF[k] = PB[k] = splat(0.f);
        }
#pragma unroll
        for (int c = 0; c < TPP; ++c) {
          X[3 + c] = xc[c].get(0);
          DO[3 + c] = dc[c].get(0);
        }
      }
      // One body per cell; xn / dn = the packed cell j + 1, consumed at the end of the body and refilled THERE with cell
      // j + 3.  The two packed buffers alternate between bodies (the loop below is unrolled by two): as `xr1 = xr2; xr2 =
      // load` the rotation is a parallel copy at the end of the loop body, the load lands in temporaries, and the copy out
      // of them waits for every load of the iteration that issued it -- the cell "two ahead" was waited for at once
      // (round 6, ISA: `s_waitcnt vmcnt(15) ... vmcnt(0)` over 16 `v_mov`s at the bottom of the loop).
      auto body = [&](int j, P (&xn)[TPP], P (&dn)[TPP]) {
        const bool tail = j == pcols, first = j == 0;
        const float e3 = tail ? m_dn : 1.f;
        f2 hcf[NH], hcb[NH];        // slot gradients of the first three steps of this body
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          if constexpr (CHAN) {
            hcf[k] = park[(tail ? k : 2 * NH + k) * 64];
            hcb[k] = park[(first ? NH + k : 2 * NH + NS + TPP - 3 + k) * 64];
          } else {
            hcf[k] = tail ? dcf_dn[k] : dcf[0];
            hcb[k] = first ? dcb_up[k] : dcb[0];
          }
        }
#pragma unroll
        for (int c = 0; c < TPP; ++c) {
          if (c < 3 || !tail) {
            // step n = TPP*j - 3 + c: pre_f of position n+3 and pre_b of position n, both from x[n .. n+3]
            f2 pf = bf, pb = bb;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
              pf = fma2(wf[k], X[c + k], pf);
              pb = fma2(wb[k], X[c + 3 - k], pb);
            }
            const f2 sgf = sigmoid2(pf), sgb = sigmoid2(pb);
            const f2 dsf = sgf * fma2(pf, 1.f - sgf, splat(1.f)), dsb = sgb * fma2(pb, 1.f - sgb, splat(1.f));
            // pooling slot of position n+3 (token c of this cell) and of position n (token c-3, or TPP-3+c of the
            // previous cell); the dense path has one slot
            const int c3 = c < 3 ? c : 0;
            const int sf = CHAN ? c : 0, sf_dn = CHAN ? c3 : 0;
            const int sb = CHAN ? (c < 3 ? TPP - 3 + c3 : c - 3) : 0, sb_up = CHAN ? c3 : 0;
            f2 cfs, cbs;
            if constexpr (CHAN) {
              cfs = c < 3 ? hcf[sf_dn] : park[(2 * NH + sf) * 64];
              cbs = c < 3 ? hcb[sb_up] : park[(2 * NH + NS + sb) * 64];
            } else {
              cfs = c < 3 ? hcf[sf_dn] : dcf[sf];
              cbs = c < 3 ? hcb[sb_up] : dcb[sb];
            }
            const float e0 = (c < 3 && first) ? m_up : 1.f;
            const f2 nf = fma2(Dfh, DO[c + 3], cfs) * dsf * e3;
            const f2 nb = fma2(Dbh, DO[c], cbs) * dsb * e0;
            PF[c + 3] = nf;
            PB[c + 3] = nb;
            if (!tail) {          // position n+3 belongs to this row
#pragma unroll
              for (int k = 0; k < CW; ++k) a_wf[k] = fma2(nf, X[c + k], a_wf[k]);
              a_bf += nf;
              a_Df = fma2(DO[c + 3], pf * sgf, a_Df);
            }
            if (c >= 3 || !first) {   // position n belongs to this row
#pragma unroll
              for (int k = 0; k < CW; ++k) a_wb[k] = fma2(nb, X[c + 3 - k], a_wb[k]);
              a_bb += nb;
              a_Db = fma2(DO[c], pb * sgb, a_Db);
              // dx[n] = sum_k wf[k] dpre_f[n+3-k] + wb[k] dpre_b[n-3+k]
              f2 dx = splat(0.f);
#pragma unroll
              for (int k = 0; k < CW; ++k) {
                dx = fma2(wf[k], PF[c + 3 - k], dx);
                dx = fma2(wb[k], PB[c + k], dx);
              }
              const int m = c < 3 ? m_prev + (TPP - 3 + c) * ts : m_cur + (c - 3) * ts;
              { const f2 dxa[1] = {dx}; P::store(bo, voff, m * tok_x, dxa); }
            }
          }
        }
        if (!tail) {
          // carry the last three positions, make cell j+1 current, fetch cell j+3
          const float m_nx = j + 1 < pcols ? 1.f : m_dn;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            X[k] = X[TPP + k];
            DO[k] = DO[TPP + k];
            PF[k] = PF[TPP + k];
            PB[k] = PB[TPP + k];
          }
#pragma unroll
          for (int c = 0; c < TPP; ++c) {
            X[3 + c] = xn[c].get(0) * m_nx;
            DO[3 + c] = dn[c].get(0);
          }
          m_prev = m_cur;
          m_cur = cell(j + 1);
          const int m3 = cell(j + 3 > pcols ? pcols : j + 3);
#pragma unroll
          for (int c = 0; c < TPP; ++c) {
            xn[c].load(bx, voff, (m3 + c * ts) * tok_x);
            dn[c].load(bd, voff, (m3 + c * ts) * tok_d);
          }
        }
      };
      for (int j = 0; j <= pcols; j += 2) {
        body(j, xr1, dr1);
        if (j + 1 <= pcols) body(j + 1, xr2, dr2);
      }
    }
  }
  flush_partials(p, smem, c0, c_lo, nch * 128, rg, RG, a_wf, a_wb, a_bf, a_bb, a_Df * 0.5f, a_Db * 0.5f);
}

template <typename T, int TPP, bool CHAN>
int launch_chan(const BwdParams& p, int nch, int rgr, int grid, int groups, size_t smem, hipStream_t st) {
  if (smem > 64 * 1024) {
    static FvOncePerDevice done;   
    if (done.first()) {
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_chan_kernel<T, TPP, CHAN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)0;     
    }
  }
  hipLaunchKernelGGL((conv_pool_bwd_chan_kernel<T, TPP, CHAN>), dim3(grid, groups), dim3(64 * nch * rgr), smem, st, p, nch, rgr);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T, int NT, bool X2>
int launch_row2(const BwdParams& p, int nch, int rgr, int grid, int groups, size_t smem, hipStream_t st) {
  if (smem > 64 * 1024) {     // opt in to > 64 KiB of dynamic LDS (once per instantiation; not a stream operation)
    static FvOncePerDevice done;
    if (done.first()) {
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_row_kernel<T, NT, X2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)0;
    }
  }
  hipLaunchKernelGGL((conv_pool_bwd_row_kernel<T, NT, X2>), dim3(grid, groups), dim3(64 * nch * rgr), smem, st, p, nch, rgr);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
template <typename T, int NT>
int launch_row(const BwdParams& p, int nch, int rgr, int grid, int groups, size_t smem, hipStream_t st) {
  return p.dxc2 ? launch_row2<T, NT, true>(p, nch, rgr, grid, groups, smem, st)
                : launch_row2<T, NT, false>(p, nch, rgr, grid, groups, smem, st);
}

}  // namespace

int fvi::conv_pool_bwd_row(const BwdParams& p, int nch, int rg, int grid, size_t smem, int dtype, hipStream_t st) {
  // nch counts 128-channel waves (a lane owns a channel pair)
  if (p.d_in != nch * 128) return FV_ERR_UNSUPPORTED;
  if ((size_t)p.geo.L * 2 * p.d_in * 4 > 0xfffff000ull) return FV_ERR_UNSUPPORTED;   // one batch element per descriptor
  if ((size_t)p.B * p.geo.rows * p.d_in * 8 > 0x7ffff000ull) return FV_ERR_UNSUPPORTED;   // pooled gradients: one descriptor, int offsets
  const bool chan8 = p.geo.tpp == 8 && p.geo.pcols >= 2;
  const bool dense8 = p.geo.tpp == 1 && p.geo.cols % 8 == 0 && p.geo.cols >= 24;      // 512 / 1024 / 2048 px grids
  const bool long_rows = chan8 || dense8;
  if (p.dxc2 && (long_rows || p.geo.tpp != 1 || (p.geo.cols != 14 && p.geo.cols != 16))) return FV_ERR_UNSUPPORTED;
  // channel groups over blockIdx.y.  Long-row kernels (201 VGPRs: 8 waves per CU): blocks of FOUR waves -- at most two waves of channels x two or four rows -- so that two
  // blocks share a CU and the dispatcher has 2-6x as many, lighter blocks to balance (six-wave blocks left a quarter of
  // the wave slots empty): FastChannelVim-S 200.5 -> 173 us, FastVim-B at 2048 px 448.7 -> 385 us
  // (profiles/r05_ab_chan_block_shapes.log)
  static const int t_groups = fv_tune("FASTVIM_BWD_CHAN_GROUPS", 0), t_rg = fv_tune("FASTVIM_BWD_CHAN_RG", 0);   // tuning hooks
  // rows live in registers: whole-row blocks of <= 512 threads (256 VGPRs per wave) for fp32 storage and 16-token rows,
  // <= 768 (168) for the bf16 14-token kernel, i.e. fewer row groups per block than the generic kernel, over the same
  // persistent grid.  The split must not leave wave slots of the CU empty (FastVim-B, 12 waves of channels: two
  // 6-wave blocks 99.5 us, one 12-wave block 73.4, three 4-wave blocks 75.6)
  const int wmax = (long_rows || dtype == FV_F32 || p.geo.cols > 14) ? 8 : 12;
  int groups = 1, nchg = nch, rgr = 1;
  if (long_rows) {
    groups = (nch + 1) / 2;
    while (nch % groups) ++groups;
    if (t_groups > 0 && nch % t_groups == 0) groups = t_groups;
    nchg = nch / groups;
    rgr = t_rg > 0 ? t_rg : 4 / nchg;
    smem = (size_t)12 * nchg * 128 * 4;           // a block accumulates its own channels only
    if (chan8) smem += (size_t)nchg * rgr * ((2 * 3 + 2 * 8) * 64) * 8;      // + the waves' parked slot gradients (11 KB per wave)
  } else {
    static const int r_groups = fv_tune("FASTVIM_BWD_ROW_GROUPS", 0), r_rg = fv_tune("FASTVIM_BWD_ROW_RG", 0);   // tuning hooks
    for (groups = 1; groups <= nch; ++groups) {
      if (nch % groups) continue;
      nchg = nch / groups;
      if (nchg > wmax) continue;
      rgr = rg < wmax / nchg ? rg : wmax / nchg;
      if (wmax % (nchg * rgr) == 0) break;
    }
    if (groups > nch) { groups = nch; nchg = 1; rgr = 1; }
    if (r_groups > 0 && nch % r_groups == 0) { groups = r_groups; nchg = nch / groups; rgr = rg < wmax / nchg ? rg : (wmax / nchg < 1 ? 1 : wmax / nchg); }
    if (r_rg > 0) rgr = r_rg;
  }
  if (chan8 || dense8) {
    static const bool chan = (fv_tune("FASTVIM_BWD_CHAN", 1) != 0);   // tuning hook
    if (!chan) return FV_ERR_UNSUPPORTED;
#define FV_CH(TT, CC) launch_chan<TT, 8, CC>(p, nchg, rgr, grid, groups, smem, st)
    if (dtype == FV_F32) return chan8 ? FV_CH(float, true) : FV_CH(float, false);
    return chan8 ? FV_CH(bf16_t, true) : FV_CH(bf16_t, false);
#undef FV_CH
  }
  if (p.geo.tpp != 1 || (p.geo.cols != 14 && p.geo.cols != 16)) return FV_ERR_UNSUPPORTED;
  static const bool wide = (fv_tune("FASTVIM_BWD_ROWK_WIDE", 1) != 0);   // tuning hook
  if (groups > 1 && !wide) return FV_ERR_UNSUPPORTED;
  if (dtype == FV_F32)
    return p.geo.cols == 14 ? launch_row<float, 14>(p, nchg, rgr, grid, groups, smem, st)
                            : launch_row<float, 16>(p, nchg, rgr, grid, groups, smem, st);
  return p.geo.cols == 14 ? launch_row<bf16_t, 14>(p, nchg, rgr, grid, groups, smem, st)
                          : launch_row<bf16_t, 16>(p, nchg, rgr, grid, groups, smem, st);
}
