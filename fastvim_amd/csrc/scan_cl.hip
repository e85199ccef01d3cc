// dt_proj + softplus + selective scan over the POOLED rows, channel-last, forward and backward,
// both scan directions in one launch.  Replaces selective_scan_cuda.fwd/bwd as used by the FastVim
// mixer (mamba_simple_faster.py:328-354, 390-410; selective_scan_interface.py:558-568, 679-696)
// plus the dt_proj matmul and its adjoint einsums (selective_scan_interface.py:515-519, 721-723).
//
// Mapping: the pooled length Lc is short (14 at 224 px, <= 128 elsewhere), so the recurrence is
// run serially in registers; parallelism comes from batch x d_inner x state-quads.  A lane owns
// 4 of the 16 states of one channel (quad q = lane & 3 owns states 4q..4q+3); the 4 lanes of a
// channel are adjacent, so the sums over states (y, du, d delta) are two DPP quad adds.  A wave
// covers 16 channels, a 256-thread block 64 channels: loads/stores of u, y, du are contiguous
// across the block.  B_t, C_t, dt_low_t are staged once per block in LDS (fp32) and read as
// broadcasts.
//
// Backward: a forward sweep checkpoints the state entering every 4-step segment (LDS for short
// pooled lengths, global scratch otherwise); segments are then walked high-to-low, recomputing the
// segment's 4 states into registers and carrying the adjoint state in registers.
// dB/dC/d dt_low need a sum over channels: a wave reduce-scatters its 16 channel lanes
// (__shfl_xor butterfly on lane bits 2..5), waves are summed through LDS in fixed order, blocks
// write per-chunk partials that fv_reduce_partials sums -- deterministic, no float atomics.
#include <stdlib.h>

#include "common.h"
#include "lane_reduce.h"

namespace {

constexpr int N = 16;        // d_state
constexpr int CPB = 64;      // channels per block (256 threads)

struct ScanClParams {
  const void* xc;        // (2, B, Lc, d_in)   pooled conv output u
  const void* xdbl;      // (2, B*Lc, R+2N)    [dt_low | B | C]
  const float* Wdt[2];   // (d_in, R)
  const float* dtb[2];   // (d_in)
  const float* Alog[2];  // (d_in, N)
  float* yc;             // (2, B, Lc, d_in)   fwd out
  const float* dyc;      // (B, Lc, d_in)      bwd in (same for both directions) or (2, B, Lc, d_in)
  size_t dyc_dir;        // elements between the two directions' dyc (0: shared)
  float* dxc;            // (2, B, Lc, d_in)   bwd out: gradient wrt u
  float* dxdbl;          // (nchunks, 2, B*Lc, R+2N) bwd out: per-chunk partial gradient wrt x_dbl
  float* ckpt;           // (2, B, nseg, d_in, N)
  float* pP;             // (B / NBB, 2, d_in*(N+R+1)) partials, per direction [dA_log | d dt_w | d dt_bias]
  int B, Lc, d_in, R;
  int NBB;               // backward: batch elements one block walks (its parameter-gradient partial covers them all)
};


template <typename T, int RQ>
struct Lane {
  int d, q, dir, b;
  bool act;
  float A2[4], Araw[4], wdt[RQ], bias;
  __device__ __forceinline__ void init(const ScanClParams& p) {
    const int tid = threadIdx.x;
    dir = blockIdx.z; b = blockIdx.y;
    q = tid & 3;
    d = blockIdx.x * CPB + (tid >> 2);
    act = d < p.d_in;
    const int dd = act ? d : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Araw[j] = -__expf(p.Alog[dir][(size_t)dd * N + q * 4 + j]);   // A = -exp(A_log) (mamba_simple_faster.py:197)
      A2[j] = Araw[j] * FV_LOG2E;
    }
#pragma unroll
    for (int i = 0; i < RQ; ++i) {
      const int r = q + 4 * i;
      wdt[i] = (r < p.R) ? p.Wdt[dir][(size_t)dd * p.R + r] : 0.f;
    }
    bias = p.dtb[dir][dd];
  }
  // softplus(dt_proj(dt_low) + bias) -- each quad lane sums its r = q, q+4, ... then quad add
  __device__ __forceinline__ float delta(const float* row) const {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < RQ; ++i) acc = fmaf(wdt[i], row[q + 4 * i], acc);   // row padded with zeros to 4*RQ
    return fv_softplus(quad_sum(acc) + bias);
  }
};

// stage x_dbl rows of this (dir, b) into LDS as fp32, row stride WP (dt_low part padded to 4*RQ)
template <typename T>
__device__ __forceinline__ void stage_dbl(const ScanClParams& p, int dir, int b, float* s_dbl, int RP) {
  const int W = p.R + 2 * N, WP = RP + 2 * N;
  const T* dbl = (const T*)p.xdbl + ((size_t)dir * p.B + b) * p.Lc * W;
  for (int e = threadIdx.x; e < p.Lc * WP; e += blockDim.x) {
    const int l = e / WP, c = e - l * WP;
    float v = 0.f;
    if (c < p.R) v = io<T>::ld(dbl + (size_t)l * W + c);
    else if (c >= RP) v = io<T>::ld(dbl + (size_t)l * W + p.R + (c - RP));
    s_dbl[e] = v;
  }
}

template <typename T, int RQ>
__global__ __launch_bounds__(256) void scan_cl_fwd_kernel(ScanClParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RP = 4 * RQ;
  const int WP = RP + 2 * N;
  Lane<T, RQ> ln;
  ln.init(p);
  stage_dbl<T>(p, ln.dir, ln.b, smem, RP);
  __syncthreads();
  const int dd = ln.act ? ln.d : 0;
  const T* u = (const T*)p.xc + ((size_t)ln.dir * p.B + ln.b) * p.Lc * p.d_in + dd;
  float* y = p.yc + ((size_t)ln.dir * p.B + ln.b) * p.Lc * p.d_in + dd;
  float st[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < p.Lc; s0 += 4) {
    // the 4 loads of the group are issued together, ahead of the arithmetic
    float uv[4];
    int lk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int step = min(s0 + k, p.Lc - 1);
      lk[k] = ln.dir ? p.Lc - 1 - step : step;       // backward direction: descending rows
      uv[k] = io<T>::ld(u + (size_t)lk[k] * p.d_in);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (s0 + k < p.Lc) {
        const float* row = smem + lk[k] * WP;
        const float dt = ln.delta(row);
        const float du = dt * uv[k];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st[j] = fmaf(fv_exp2(dt * ln.A2[j]), st[j], du * row[RP + ln.q * 4 + j]);
          acc = fmaf(row[RP + N + ln.q * 4 + j], st[j], acc);
        }
        acc = quad_sum(acc);
        if (ln.act && ln.q == 0) y[(size_t)lk[k] * p.d_in] = acc;
      }
    }
  }
}

// Backward.  Segments of KS = 4 steps: the states entering every segment are checkpointed by a
// forward sweep (in LDS when the pooled length is short, else in global scratch), then segments are
// walked high-to-low: recompute the 4 states (registers), run the adjoint, reduce over channels.
template <typename T, int RQ, int PV, bool CK_LDS, bool DTC>
__global__ __launch_bounds__(256) void scan_cl_bwd_kernel(ScanClParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KS = 4;
  constexpr int RP = 4 * RQ;
  constexpr int Q = PV / 16;                // values a lane holds after the butterfly
  constexpr int NWV = 4;                    // waves per block
  const int WP = RP + 2 * N, W = p.R + 2 * N;
  const int nseg = (p.Lc + KS - 1) / KS;
  float* s_dbl = smem;                      // Lc * WP
  float* s_part = s_dbl + p.Lc * WP;        // KS * NWV * 4 * PV   [k][wave][q][value]
  // DTC (short pooled lengths): softplus(dt_proj) of every (row, channel) is computed once into LDS; for long
  // sequences that table would cost occupancy (Lc * 256 B per block), so delta is recomputed where it is used
  float* s_dt = s_part + KS * NWV * 4 * PV; // Lc * CPB (DTC only)
  float* s_ck = s_dt + (DTC ? p.Lc * CPB : 0);   // nseg * 256 * 4 (CK_LDS only)
  Lane<T, RQ> ln;
  ln.init(p);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float dA[4] = {0.f, 0.f, 0.f, 0.f}, dW[RQ], dbias = 0.f;     // parameter gradients: summed over this block's batch elements
#pragma unroll
  for (int i = 0; i < RQ; ++i) dW[i] = 0.f;
  for (int bi = 0; bi < p.NBB; ++bi) {
  ln.b = blockIdx.y * p.NBB + bi;
  if (bi) __syncthreads();                  // the previous element's readers are done with the LDS stage
  stage_dbl<T>(p, ln.dir, ln.b, s_dbl, RP);
  __syncthreads();
  if constexpr (DTC) {
    for (int l = 0; l < p.Lc; ++l) {
      const float dt = ln.delta(s_dbl + l * WP);
      if (ln.q == 0) s_dt[l * CPB + (tid >> 2)] = dt;
    }
    __syncthreads();
  }
  const int dd = ln.act ? ln.d : 0;
  const T* u = (const T*)p.xc + ((size_t)ln.dir * p.B + ln.b) * p.Lc * p.d_in + dd;
  const float* gy = p.dyc + (size_t)ln.dir * p.dyc_dir + (size_t)ln.b * p.Lc * p.d_in + dd;
  float* ckg = p.ckpt + (((size_t)ln.dir * p.B + ln.b) * nseg * p.d_in + dd) * N + ln.q * 4;
  const size_t ck_seg = (size_t)p.d_in * N;

  // ---- forward sweep: checkpoint the state entering every segment but the first
  {
    float st[4] = {0.f, 0.f, 0.f, 0.f};
    for (int seg = 0; seg + 1 < nseg; ++seg) {
      float uv[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int step = seg * KS + k;
        const int l = ln.dir ? p.Lc - 1 - step : step;
        uv[k] = io<T>::ld(u + (size_t)l * p.d_in);
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int step = seg * KS + k;
        const int l = ln.dir ? p.Lc - 1 - step : step;
        const float* row = s_dbl + l * WP;
        const float dt = DTC ? s_dt[l * CPB + (tid >> 2)] : ln.delta(row);
        const float du = dt * uv[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) st[j] = fmaf(fv_exp2(dt * ln.A2[j]), st[j], du * row[RP + ln.q * 4 + j]);
      }
      if (CK_LDS) {
        *reinterpret_cast<float4*>(s_ck + ((size_t)(seg + 1) * 256 + tid) * 4) = make_float4(st[0], st[1], st[2], st[3]);
      } else if (ln.act) {
        *reinterpret_cast<float4*>(ckg + (size_t)(seg + 1) * ck_seg) = make_float4(st[0], st[1], st[2], st[3]);
      }
    }
  }

  float dxa[4] = {0.f, 0.f, 0.f, 0.f};

  for (int seg = nseg - 1; seg >= 0; --seg) {
    const int s0 = seg * KS;
    const int ns = min(KS, p.Lc - s0);
    // loads of the whole segment first (clamped rows; masked below), then the arithmetic
    float uv[KS], gq[KS];
    int lk[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int step = min(s0 + k, p.Lc - 1);
      lk[k] = ln.dir ? p.Lc - 1 - step : step;
      uv[k] = io<T>::ld(u + (size_t)lk[k] * p.d_in);
      gq[k] = gy[(size_t)lk[k] * p.d_in];
    }
    float cur[4] = {0.f, 0.f, 0.f, 0.f};
    if (seg > 0) {
      float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (CK_LDS) c4 = *reinterpret_cast<const float4*>(s_ck + ((size_t)seg * 256 + tid) * 4);
      else if (ln.act) c4 = *reinterpret_cast<const float4*>(ckg + (size_t)seg * ck_seg);
      cur[0] = c4.x; cur[1] = c4.y; cur[2] = c4.z; cur[3] = c4.w;
    }
    float xs[KS][4], aq[KS][4], dtv[KS];   // states, decay factors exp(delta*A) (reused by the adjoint), deltas
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float* row = s_dbl + lk[k] * WP;
      const bool on = k < ns && ln.act;
      dtv[k] = on ? (DTC ? s_dt[lk[k] * CPB + (tid >> 2)] : ln.delta(row)) : 0.f;      // delta = 0: a = 1, b = 0 -> identity step
      if (!on) gq[k] = 0.f;
      const float du = dtv[k] * uv[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        aq[k][j] = fv_exp2(dtv[k] * ln.A2[j]);
        cur[j] = fmaf(aq[k][j], cur[j], du * row[RP + ln.q * 4 + j]);
        xs[k][j] = cur[j];
      }
    }
    // adjoint recurrence, high-to-low
#pragma unroll
    for (int k = KS - 1; k >= 0; --k) {
      if (k < ns) {          // uniform across the block
        const float* row = s_dbl + lk[k] * WP;
        float vals[PV];
#pragma unroll
        for (int e = 0; e < PV; ++e) vals[e] = 0.f;
        float du_acc = 0.f, ddt_acc = 0.f;
        const float dtu = dtv[k] * uv[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float Bn = row[RP + ln.q * 4 + j], Cn = row[RP + N + ln.q * 4 + j];
          const float a = aq[k][j];
          const float dx = fmaf(gq[k], Cn, dxa[j]);
          const float ax = xs[k][j] - dtu * Bn;               // a_t * x_{t-1}
          du_acc = fmaf(dx, Bn, du_acc);
          ddt_acc += dx * fmaf(ln.Araw[j], ax, Bn * uv[k]);
          dA[j] = fmaf(dx * dtv[k], ax, dA[j]);
          vals[j] = dx * dtu;                                  // dB[4q+j]
          vals[4 + j] = gq[k] * xs[k][j];                      // dC[4q+j]
          dxa[j] = a * dx;
        }
        du_acc = quad_sum(du_acc);
        ddt_acc = quad_sum(ddt_acc);
        // d softplus: sigmoid(raw) = 1 - exp(-softplus(raw))
        const float ddraw = ln.act ? ddt_acc * (1.f - __expf(-dtv[k])) : 0.f;
        dbias += ddraw;
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
          dW[i] = fmaf(ddraw, row[ln.q + 4 * i], dW[i]);
          vals[8 + i] = ddraw * ln.wdt[i];                     // d dt_low[q + 4i]
        }
        if (ln.act && ln.q == 0)
          p.dxc[(((size_t)ln.dir * p.B + ln.b) * p.Lc + lk[k]) * p.d_in + ln.d] = dtv[k] * du_acc;
        chan_reduce_scatter<PV>(vals, lane);
        const int c = lane >> 2;
#pragma unroll
        for (int e = 0; e < Q; ++e) s_part[((k * NWV + wv) * 4 + ln.q) * PV + c * Q + e] = vals[e];
      }
    }
    __syncthreads();
    // sum the 4 waves in fixed order and scatter to the x_dbl column layout [dt_low | B | C]
    for (int e = tid; e < ns * 4 * PV; e += blockDim.x) {
      const int k = e / (4 * PV), rem = e - k * 4 * PV;
      const int qq = rem / PV, v = rem - qq * PV;
      int col = -1;
      if (v < 4) col = p.R + qq * 4 + v;
      else if (v < 8) col = p.R + N + qq * 4 + (v - 4);
      else if (v < 8 + RQ && qq + 4 * (v - 8) < p.R) col = qq + 4 * (v - 8);
      if (col >= 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += s_part[((k * NWV + w) * 4 + qq) * PV + v];
        const int step = s0 + k;
        const int l = ln.dir ? p.Lc - 1 - step : step;
        p.dxdbl[(((size_t)blockIdx.x * 2 + ln.dir) * p.B + ln.b) * p.Lc * W + (size_t)l * W + col] = t;
      }
    }
    __syncthreads();
  }
  }   // batch elements of this block
  if (ln.act) {
    // one partial row per block row (NBB batch elements), per direction: [dA_log (d_in*N) | d dt_w (d_in*R) | d dt_bias (d_in)]
    const size_t per_dir = (size_t)p.d_in * (N + p.R + 1);
    float* base = p.pP + ((size_t)blockIdx.y * 2 + ln.dir) * per_dir;
#pragma unroll
    for (int j = 0; j < 4; ++j) base[(size_t)ln.d * N + ln.q * 4 + j] = dA[j] * ln.Araw[j];   // dA_log = dA * A
#pragma unroll
    for (int i = 0; i < RQ; ++i)
      if (ln.q + 4 * i < p.R) base[(size_t)p.d_in * N + (size_t)ln.d * p.R + ln.q + 4 * i] = dW[i];
    if (ln.q == 0) base[(size_t)p.d_in * (N + p.R) + ln.d] = dbias;                           // identical in all 4 lanes
  }
}


// ------------------------------------------------------------------------------------------------------------
// Backward for SHORT pooled lengths (Lc <= 16: the 224 / 256 px grids, BASELINE configs 2 and 3).
// Everything of a (batch element, direction, 192-channel chunk) lives in registers: the forward recurrence is run
// ONCE and its 14 x 4 states per lane are kept (no checkpoint sweep, no segment recompute, no barrier inside the
// time loop); the decay factors are re-derived in the adjoint sweep (one v_exp_f32 each) because keeping them too
// would cost a wave of occupancy.  Per-channel values are spread over the four state-quad lanes of a channel IN TIME:
// lane q holds u, dy, delta and sigmoid(delta_raw) of the steps s = q (mod 4) and the others read them as DPP
// quad-broadcast operands, so the loads, the softplus and the sigmoid of a step are done once per channel, not four
// times.  One 768-thread workgroup (12 waves, 3 per SIMD, one workgroup per CU) walks NBB batch elements of its
// chunk: 256 workgroups at FastVim-T bs 128 -- exactly one round -- and the parameter-gradient partials shrink by
// NBB.  d x_dbl needs a sum over channels: in-wave reduce-scatter as in the long kernel, then ONE fixed-order sum of
// the 12 waves through LDS per batch element (two barriers per element in all).
constexpr int SH_CH = 192, SH_THREADS = 768, SH_NWV = 12;

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {      // value of quad lane K, in all four lanes (DPP quad_perm)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_pick(float v, int k) {  // k is a compile-time constant after unrolling
  switch (k & 3) {
    case 0: return quad_bcast<0>(v);
    case 1: return quad_bcast<1>(v);
    case 2: return quad_bcast<2>(v);
    default: return quad_bcast<3>(v);
  }
}

template <typename T, int RQ, int PV, int LCT, bool EXACT>      // EXACT: Lc == LCT (the 14- and 16-row grids)
__global__ __launch_bounds__(SH_THREADS) void scan_cl_bwd_short_kernel(ScanClParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RQP = (RQ + 3) / 4 * 4;     // dt_low part of a staged row: [q][RQP] (zero padded), 16-byte groups
  constexpr int WP = 4 * RQP + 2 * N;       // staged row: [dt_low by quad | B | C]
  constexpr int NG = (LCT + 3) / 4;         // step groups of 4 (one step per quad lane)
  constexpr int Q = PV / 16;
  float* s_dbl = smem;                      // LCT * WP, rows in SCAN order (row s = step s): compile-time LDS offsets
  float* s_part = smem + LCT * WP;          // LCT * SH_NWV * 4 * PV   [step][wave][q][value]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = tid & 3;
  const int dir = blockIdx.z;
  const int d = blockIdx.x * SH_CH + (tid >> 2);
  const bool act = d < p.d_in;
  const int dd = act ? d : 0;
  const int W = p.R + 2 * N;
  const int Lc = EXACT ? LCT : p.Lc;
  float A2[4], Araw[4], wdt[RQ];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    Araw[j] = -__expf(p.Alog[dir][(size_t)dd * N + q * 4 + j]);
    A2[j] = Araw[j] * FV_LOG2E;
  }
#pragma unroll
  for (int i = 0; i < RQ; ++i) {
    const int r = q + 4 * i;
    wdt[i] = (r < p.R) ? p.Wdt[dir][(size_t)dd * p.R + r] : 0.f;
  }
  const float bias = p.dtb[dir][dd];
  float dA[4] = {0.f, 0.f, 0.f, 0.f}, dW[RQ], dbias = 0.f;
#pragma unroll
  for (int i = 0; i < RQ; ++i) dW[i] = 0.f;
  const float* my_dl = s_dbl + q * RQP;                 // this quad lane's dt_low group of row 0
  const float* my_bc = s_dbl + 4 * RQP + q * 4;         // this quad lane's B states of row 0 (C: + N)
  float* my_part = s_part + (wv * 4 + q) * PV + (lane >> 2) * Q;

  for (int bi = 0; bi < p.NBB; ++bi) {
    const int b = blockIdx.y * p.NBB + bi;
    const size_t bd = ((size_t)dir * p.B + b) * Lc;
    const T* u = (const T*)p.xc + bd * p.d_in + dd;
    const float* gy = p.dyc + (size_t)dir * p.dyc_dir + (size_t)b * Lc * p.d_in + dd;
    // this lane's share of the per-channel inputs: steps s = 4i + q (requested before the staging barrier)
    float ur[NG], gr[NG];
    int roff[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int s = 4 * i + q, sc = min(s, Lc - 1), l = dir ? Lc - 1 - sc : sc;
      roff[i] = l * p.d_in;
      ur[i] = io<T>::ld(u + roff[i]);
      const float gv = gy[roff[i]];
      gr[i] = (act && s < Lc) ? gv : 0.f;
    }
    // stage the x_dbl rows of (dir, b) in scan order: fp32, dt_low regrouped by quad lane, rows past Lc zero
    {
      const T* dbl = (const T*)p.xdbl + bd * W;
      for (int e = tid; e < LCT * WP; e += SH_THREADS) {
        const int srow = e / WP, c = e - srow * WP;
        const int l = dir ? Lc - 1 - srow : srow;
        float v = 0.f;
        if (EXACT || srow < Lc) {
          if (c < 4 * RQP) {
            const int qq = c / RQP, i = c - qq * RQP, r = qq + 4 * i;
            if (i < RQ && r < p.R) v = io<T>::ld(dbl + (size_t)l * W + r);
          } else {
            v = io<T>::ld(dbl + (size_t)l * W + p.R + (c - 4 * RQP));
          }
        }
        s_dbl[e] = v;
      }
    }
    __syncthreads();      // rows staged; the previous element's readers of s_part are done as well

    // ---- delta = softplus(dt_proj(dt_low) + bias) and its derivative, one step per quad lane and group
    float dtr[NG], sgr[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      float mine = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (4 * i + k < LCT) {
          const float* row = my_dl + (4 * i + k) * WP;
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < RQ; ++j) acc = fmaf(wdt[j], row[j], acc);
          acc = quad_sum(acc);
          mine = (q == k) ? acc : mine;
        }
      }
      const bool on = act && (4 * i + q < Lc);
      const float dtq = on ? fv_softplus(mine + bias) : 0.f;     // delta = 0: identity step (a = 1, b = 0)
      dtr[i] = dtq;
      sgr[i] = 1.f - __expf(-dtq);                                // sigmoid(raw) = 1 - exp(-softplus(raw)); 0 when off
    }

    // ---- forward recurrence, states kept
    float xs[LCT][4];
    {
      float st[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < LCT; ++s) {
        const float4 Bv = *reinterpret_cast<const float4*>(my_bc + s * WP);
        const float Bn[4] = {Bv.x, Bv.y, Bv.z, Bv.w};
        const float dt = quad_pick(dtr[s >> 2], s), uu = quad_pick(ur[s >> 2], s);
        const float dtu = dt * uu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st[j] = fmaf(fv_exp2(dt * A2[j]), st[j], dtu * Bn[j]);
          xs[s][j] = st[j];
        }
      }
    }

    // ---- adjoint sweep, high to low
    float dxa[4] = {0.f, 0.f, 0.f, 0.f};
    float dxr[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) dxr[i] = 0.f;
#pragma unroll
    for (int s = LCT - 1; s >= 0; --s) {
      if (s < p.Lc) {         // uniform; a real branch also in the EXACT build: one basic block per step keeps the
                              // scheduler from hoisting every step's LDS reads to the top (it spilled 145 VGPRs)
        const float4 Bv = *reinterpret_cast<const float4*>(my_bc + s * WP);
        const float4 Cv = *reinterpret_cast<const float4*>(my_bc + s * WP + N);
        const float Bn[4] = {Bv.x, Bv.y, Bv.z, Bv.w}, Cn[4] = {Cv.x, Cv.y, Cv.z, Cv.w};
        const float dt = quad_pick(dtr[s >> 2], s), uu = quad_pick(ur[s >> 2], s);
        const float g = quad_pick(gr[s >> 2], s), sg = quad_pick(sgr[s >> 2], s);
        const float dtu = dt * uu;
        float vals[PV];
#pragma unroll
        for (int e = 0; e < PV; ++e) vals[e] = 0.f;
        float du_acc = 0.f, dd_acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = fv_exp2(dt * A2[j]);
          const float dx = fmaf(g, Cn[j], dxa[j]);
          const float pj = s > 0 ? dx * (a * xs[s > 0 ? s - 1 : 0][j]) : 0.f;     // dx * a_t * x_{t-1}
          du_acc = fmaf(dx, Bn[j], du_acc);
          dd_acc = fmaf(Araw[j], pj, dd_acc);
          dA[j] = fmaf(dt, pj, dA[j]);
          vals[j] = dx * dtu;                                   // dB[4q+j]
          vals[4 + j] = g * xs[s][j];                           // dC[4q+j]
          dxa[j] = a * dx;
        }
        du_acc = quad_sum(du_acc);
        dd_acc = quad_sum(dd_acc);
        // d delta = sum_n dx (B u + A a x_prev) = u * sum_n dx B + sum_n A dx a x_prev;  through the softplus: * sigmoid
        const float ddraw = fmaf(uu, du_acc, dd_acc) * sg;
        dbias += ddraw;
        const float* dl = my_dl + s * WP;
#pragma unroll
        for (int i = 0; i < RQ; ++i) {
          dW[i] = fmaf(ddraw, dl[i], dW[i]);
          vals[8 + i] = ddraw * wdt[i];                         // d dt_low[q + 4i]
        }
        dxr[s >> 2] = (q == (s & 3)) ? dt * du_acc : dxr[s >> 2];      // d u of step s: kept by the lane that loaded u_s
        chan_reduce_scatter<PV>(vals, lane);
#pragma unroll
        for (int e = 0; e < Q; ++e) my_part[s * (SH_NWV * 4 * PV) + e] = vals[e];
      }
    }
#pragma unroll
    for (int i = 0; i < NG; ++i)
      if (act && 4 * i + q < Lc) p.dxc[bd * p.d_in + roff[i] + d] = dxr[i];
    __syncthreads();
    // sum the 12 waves in fixed order and scatter to the x_dbl column layout [dt_low | B | C]
    for (int e = tid; e < Lc * 4 * PV; e += SH_THREADS) {
      const int s = e / (4 * PV), rem = e - s * 4 * PV;
      const int qq = rem / PV, v = rem - qq * PV;
      int col = -1;
      if (v < 4) col = p.R + qq * 4 + v;
      else if (v < 8) col = p.R + N + qq * 4 + (v - 4);
      else if (v < 8 + RQ && qq + 4 * (v - 8) < p.R) col = qq + 4 * (v - 8);
      if (col >= 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SH_NWV; ++w) t += s_part[((s * SH_NWV + w) * 4 + qq) * PV + v];
        const int l = dir ? Lc - 1 - s : s;
        p.dxdbl[(((size_t)blockIdx.x * 2 + dir) * p.B + b) * Lc * W + (size_t)l * W + col] = t;
      }
    }
  }   // batch elements of this block
  if (act) {
    const size_t per_dir = (size_t)p.d_in * (N + p.R + 1);
    float* base = p.pP + ((size_t)blockIdx.y * 2 + dir) * per_dir;
#pragma unroll
    for (int j = 0; j < 4; ++j) base[(size_t)d * N + q * 4 + j] = dA[j] * Araw[j];            // dA_log = dA * A
#pragma unroll
    for (int i = 0; i < RQ; ++i)
      if (q + 4 * i < p.R) base[(size_t)p.d_in * N + (size_t)d * p.R + q + 4 * i] = dW[i];
    if (q == 0) base[(size_t)p.d_in * (N + p.R) + d] = dbias;           // identical in the four lanes of a channel
  }
}

int rq_of(int R) { return (R + 3) / 4; }

}  // namespace

extern "C" int fv_mixer_scan_fwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                                 const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                                 const float* A_log_b, float* yc, int batch, int Lc, int d_inner, int dt_rank,
                                 int d_state, int dtype, fv_stream_t stream) {
  FV_CHECK(batch > 0 && Lc > 0 && d_inner > 0 && dt_rank > 0, "mixer_scan_fwd: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer_scan_fwd: dtype must be fp32 or bf16");
  FV_CHECK(d_state == N, "mixer_scan_fwd: only d_state == 16 is built (got %d)", d_state);
  FV_CHECK(dt_rank <= 96, "mixer_scan_fwd: dt_rank %d > 96", dt_rank);
  FV_CHECK(xc && x_dbl && dt_w && dt_bias && A_log && dt_w_b && dt_bias_b && A_log_b && yc,
           "mixer_scan_fwd: null pointer");
  ScanClParams p{};
  p.xc = xc; p.xdbl = x_dbl; p.yc = yc;
  p.Wdt[0] = dt_w; p.Wdt[1] = dt_w_b; p.dtb[0] = dt_bias; p.dtb[1] = dt_bias_b;
  p.Alog[0] = A_log; p.Alog[1] = A_log_b;
  p.B = batch; p.Lc = Lc; p.d_in = d_inner; p.R = dt_rank;
  const int RQ = rq_of(dt_rank);
  dim3 grid(fv_cdiv(d_inner, CPB), batch, 2), block(256);
  hipStream_t st = (hipStream_t)stream;
#define FV_F(TT, RQQ)                                                                        \
  do {                                                                                       \
    size_t smem = (size_t)Lc * (4 * RQQ + 2 * N) * 4;                                        \
    FV_CHECK(smem <= 64 * 1024, "mixer_scan_fwd: pooled length %d too long for the LDS stage", Lc); \
    hipLaunchKernelGGL((scan_cl_fwd_kernel<TT, RQQ>), grid, block, smem, st, p);             \
  } while (0)
#define FV_FD(TT)                                                                            \
  do {                                                                                       \
    if (RQ <= 3) FV_F(TT, 3); else if (RQ <= 6) FV_F(TT, 6); else if (RQ <= 12) FV_F(TT, 12); else FV_F(TT, 24); \
  } while (0)
  if (dtype == FV_F32) FV_FD(float); else FV_FD(bf16_t);
#undef FV_FD
#undef FV_F
  FV_LAUNCH_CHECK();
  return FV_OK;
}

// pooled lengths up to 16 take the register-resident kernel (192-channel workgroups)
static bool bwd_short(int Lc) {
  static const int off = getenv("FASTVIM_SCAN_SHORT") ? atoi(getenv("FASTVIM_SCAN_SHORT")) == 0 : 0;   // A/B hook
  return Lc <= 16 && !off;
}
extern "C" int fv_mixer_scan_bwd_chunks(int d_inner, int Lc) { return fv_cdiv(d_inner, bwd_short(Lc) ? SH_CH : CPB); }

// A block can walk several batch elements (fewer, longer blocks; parameter-gradient partials shrink by the same
// factor).  Measured on FastVim-T: 2 per block 49.0 us vs 47-48 us, 4 per block 65 us -- so one, unless forced.
static int scan_bwd_nbb(int batch, int Lc) {
  static const int force = getenv("FASTVIM_SCAN_NBB") ? atoi(getenv("FASTVIM_SCAN_NBB")) : 0;   // tuning hook
  if (force > 0 && batch % force == 0) return force;
  return (bwd_short(Lc) && batch % 2 == 0) ? 2 : 1;
}
extern "C" int fv_mixer_scan_bwd_partials(int batch, int Lc) { return batch / scan_bwd_nbb(batch, Lc); }

static bool ck_in_lds(int Lc) { return ((Lc + 3) / 4) * 4096 + Lc * CPB * 4 <= 32 * 1024; }

extern "C" size_t fv_mixer_scan_bwd_ckpt_floats(int batch, int Lc, int d_inner, int d_state) {
  if (bwd_short(Lc) || ck_in_lds(Lc)) return 0;
  return (size_t)2 * batch * ((Lc + 3) / 4) * d_inner * d_state;
}

extern "C" int fv_mixer_scan_bwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                                 const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                                 const float* A_log_b, const float* dyc, float* dxc, float* dx_dbl, float* ckpt,
                                 float* partials, int batch, int Lc, int d_inner, int dt_rank, int d_state,
                                 int dtype, fv_stream_t stream) {
  return fv_mixer_scan_bwd_dir(xc, x_dbl, dt_w, dt_bias, A_log, dt_w_b, dt_bias_b, A_log_b, dyc, 0, dxc, dx_dbl, ckpt,
                               partials, batch, Lc, d_inner, dt_rank, d_state, dtype, stream);
}

extern "C" int fv_mixer_scan_bwd_dir(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                                     const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                                     const float* A_log_b, const float* dyc, int dyc_per_direction, float* dxc,
                                     float* dx_dbl, float* ckpt, float* partials, int batch, int Lc, int d_inner,
                                     int dt_rank, int d_state, int dtype, fv_stream_t stream) {
  FV_CHECK(batch > 0 && Lc > 0 && d_inner > 0 && dt_rank > 0, "mixer_scan_bwd: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer_scan_bwd: dtype must be fp32 or bf16");
  FV_CHECK(d_state == N, "mixer_scan_bwd: only d_state == 16 is built (got %d)", d_state);
  FV_CHECK(dt_rank <= 96, "mixer_scan_bwd: dt_rank %d > 96", dt_rank);
  FV_CHECK(xc && x_dbl && dt_w && dt_bias && A_log && dt_w_b && dt_bias_b && A_log_b && dyc && dxc && dx_dbl &&
               partials && (ckpt || ck_in_lds(Lc)), "mixer_scan_bwd: null pointer");
  ScanClParams p{};
  p.xc = xc; p.xdbl = x_dbl; p.dyc = dyc; p.dxc = dxc; p.dxdbl = dx_dbl; p.ckpt = ckpt; p.pP = partials;
  p.dyc_dir = dyc_per_direction ? (size_t)batch * Lc * d_inner : 0;
  p.Wdt[0] = dt_w; p.Wdt[1] = dt_w_b; p.dtb[0] = dt_bias; p.dtb[1] = dt_bias_b;
  p.Alog[0] = A_log; p.Alog[1] = A_log_b;
  p.B = batch; p.Lc = Lc; p.d_in = d_inner; p.R = dt_rank;
  const int RQ = rq_of(dt_rank);
  const bool ckl = ck_in_lds(Lc);
  p.NBB = scan_bwd_nbb(batch, Lc);
  hipStream_t st = (hipStream_t)stream;
  if (bwd_short(Lc)) {
    dim3 sgrid(fv_cdiv(d_inner, SH_CH), batch / p.NBB, 2), sblock(SH_THREADS);
#define FV_S(TT, RQQ, PVV, LCC, EXX)                                                            \
  do {                                                                                       \
    constexpr int RQP_ = (RQQ + 3) / 4 * 4;                                                  \
    size_t smem = ((size_t)LCC * (4 * RQP_ + 2 * N) + (size_t)LCC * SH_NWV * 4 * PVV) * 4;   \
    static bool done = false;                                                                \
    if (!done && smem > 64 * 1024) {                                                         \
      (void)hipFuncSetAttribute((const void*)scan_cl_bwd_short_kernel<TT, RQQ, PVV, LCC, EXX>, \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      done = true;                                                                           \
    }                                                                                        \
    hipLaunchKernelGGL((scan_cl_bwd_short_kernel<TT, RQQ, PVV, LCC, EXX>), sgrid, sblock, smem, st, p); \
  } while (0)
#define FV_SL(TT, RQQ, PVV)                                                                  \
  do {                                                                                       \
    if (Lc == 14) FV_S(TT, RQQ, PVV, 14, true); else if (Lc < 14) FV_S(TT, RQQ, PVV, 14, false); \
    else if (Lc == 16) FV_S(TT, RQQ, PVV, 16, true); else FV_S(TT, RQQ, PVV, 16, false);     \
  } while (0)
#define FV_SD(TT)                                                                            \
  do {                                                                                       \
    if (RQ <= 3) FV_SL(TT, 3, 16); else if (RQ <= 6) FV_SL(TT, 6, 16);                       \
    else if (RQ <= 12) FV_SL(TT, 12, 32); else FV_SL(TT, 24, 32);                            \
  } while (0)
    if (dtype == FV_F32) FV_SD(float); else FV_SD(bf16_t);
#undef FV_SD
#undef FV_SL
#undef FV_S
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  dim3 grid(fv_cdiv(d_inner, CPB), batch / p.NBB, 2), block(256);
#define FV_B(TT, RQQ, PVV, CKK, DTT)                                                         \
  do {                                                                                       \
    size_t smem = ((size_t)Lc * (4 * RQQ + 2 * N) + (size_t)4 * 4 * 4 * PVV + (DTT ? (size_t)Lc * CPB : 0) + \
                   (CKK ? (size_t)((Lc + 3) / 4) * 1024 : 0)) * 4;                           \
    FV_CHECK(smem <= 160 * 1024, "mixer_scan_bwd: pooled length %d too long for the LDS stage", Lc); \
    if (smem > 64 * 1024) {                                                                  \
      static bool done = false;                                                              \
      if (!done) {                                                                           \
        (void)hipFuncSetAttribute((const void*)scan_cl_bwd_kernel<TT, RQQ, PVV, CKK, DTT>,   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        done = true;                                                                         \
      }                                                                                      \
    }                                                                                        \
    hipLaunchKernelGGL((scan_cl_bwd_kernel<TT, RQQ, PVV, CKK, DTT>), grid, block, smem, st, p); \
  } while (0)
#define FV_BK(TT, RQQ, PVV)                                                                  \
  do {                                                                                       \
    if (ckl) FV_B(TT, RQQ, PVV, true, true); else FV_B(TT, RQQ, PVV, false, false);          \
  } while (0)
#define FV_BD(TT)                                                                            \
  do {                                                                                       \
    if (RQ <= 3) FV_BK(TT, 3, 16); else if (RQ <= 6) FV_BK(TT, 6, 16);                       \
    else if (RQ <= 12) FV_BK(TT, 12, 32); else FV_BK(TT, 24, 32);                            \
  } while (0)
  if (dtype == FV_F32) FV_BD(float); else FV_BD(bf16_t);
#undef FV_BD
#undef FV_BK
#undef FV_B
  FV_LAUNCH_CHECK();
  return FV_OK;
}
