// hipcc-flags: -fgpu-flush-denormals-to-zero
// Backward kernels of the fused FastVim mixer "middle" (channel-last).  Hand-written adjoint of
// csrc/mixer_fwd.hip; replaces the autograd graph of mamba_simple_faster.py:270-444 and the
// hand-written backward of FastVim_MambaInnerFnNoOutProj_withoutZ
// (mamba_ssm/ops/selective_scan_interface.py:607-776):
//
//   fv_mixer_combine_bwd   : d(gate), d(LayerNorm) with xhat rebuilt from the saved skip/yc/mean/rstd -> dz, do, per-row pooled
//                            dyc = 0.5*sum_j do, partials of dLN.weight/bias.
//   (fv_mixer_scan_bwd, the adjoint of dt_proj + scan, lives in scan_cl.hip)
//   fv_mixer_conv_pool_bwd : adjoint of the D-skip, mean-pool, SiLU and both depthwise convs -> dx,
//                            and partials of the conv weight/bias and D, D_b gradients.
//   fv_reduce_partials     : fixed-order sum of per-block partials (no float atomics anywhere).
#include <stdlib.h>

#include "mixer_common.h"

namespace {

using fvi::BwdParams;

constexpr int RGMAX = 4;   // a block walks up to RGMAX pooling rows concurrently (one per row group) and emits ONE partial

// ------------------------------------------------------------------ LayerNorm + gate backward
// Reads dg, z, skip once, writes dz and d_o once: 5 full-length tensors of traffic, no conv recompute
// (xhat = ((yc_f + yc_b + skip)/2 - mean) * rstd is two FMAs from what the forward saved).  Block = RG row groups x NCH waves; a row group walks one pooling row.
template <typename T, int VEC, int TT, bool TP>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void combine_bwd_kernel(BwdParams p, int nch, int RG) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: row/token index math stays on the scalar unit
  const int rg = wv / nch, cw = wv - rg * nch;          // row group, channel-chunk wave
  const int c0 = (cw * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  float* s_red = smem;                                   // RGMAX * 2 * TT * 16 (cross-wave LN sums)
  float* s_acc = smem + RGMAX * 2 * TT * 16;             // 2 * d_in
  float* s_dyc = s_acc + 2 * p.d_in;                     // tpp > 1: thread-private [slot][thread][VEC] pooled sums
  const int tpp = TP ? g.tpp : 1, nthr = blockDim.x;
  float lw[VEC], lb[VEC], a_lw[VEC], a_lb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    lw[v] = act && p.use_norm ? p.lnw[c0 + v] : 1.f;
    lb[v] = act && p.use_norm ? p.lnb[c0 + v] : 0.f;
    a_lw[v] = a_lb[v] = 0.f;
  }
  const float inv_d = 1.f / (float)p.d_in;
  const int nrows = p.B * g.rows;
  const int nit = (nrows + gridDim.x * RG - 1) / (gridDim.x * RG);
  for (int it = 0; it < nit; ++it) {
    const int row = (it * gridDim.x + blockIdx.x) * RG + rg;
    const bool rv = row < nrows;                         // uniform per row group
    const int b = rv ? row / g.rows : 0, i = rv ? row - b * g.rows : 0;
    const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
    const T* dg_b = (const T*)p.dg + (size_t)b * g.L * p.d_in;
    const T* xh_b = (const T*)p.skip + (size_t)b * g.L * p.d_in;
    const size_t ydir = (size_t)p.B * g.rows * tpp * p.d_in;
    const float* yc_r = p.yc + (size_t)(rv ? row : 0) * tpp * p.d_in + (act ? c0 : 0);
    float ysum[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) ysum[v] = (!TP && rv && act) ? yc_r[v] + yc_r[ydir + v] : 0.f;
    T* dxz_b = (T*)p.dxz + (size_t)b * g.L * 2 * p.d_in;
    T* dob_b = (T*)p.dob + (size_t)b * g.L * p.d_in;
    float dyc_acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) dyc_acc[v] = 0.f;
    const bool nopool = TP && g.pcols == 1;     // un-pooled (Vim) form: dyc is written per token, no accumulators
    if constexpr (TP)
      if (!nopool)
        for (int c = 0; c < tpp; ++c)
#pragma unroll
          for (int v = 0; v < VEC; ++v) s_dyc[(c * nthr + threadIdx.x) * VEC + v] = 0.f;
    // software pipeline: the (packed) loads of token group j0+TT are in flight while group j0 is processed
    RawVec<T, VEC> n_dg[TT], n_z[TT], n_xh[TT];
    auto fetch = [&](int j0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (rv && act && j0 + t < g.cols) {
          const int m = tok_mem<TP>(g, i * g.cols + j0 + t);
          n_dg[t].load(dg_b + (size_t)m * p.d_in + c0);
          n_z[t].load(xz_b + (size_t)m * 2 * p.d_in + p.d_in + c0);
          n_xh[t].load(xh_b + (size_t)m * p.d_in + c0);
        } else {
          n_dg[t].zero(); n_z[t].zero(); n_xh[t].zero();
        }
      }
    };
    fetch(0);
    for (int j0 = 0; j0 < g.cols; j0 += TT) {
      float xh[TT][VEC], dxh[TT][VEC], c1[TT], c2[TT], rs[TT];
      float dgq[TT][VEC], zq[TT][VEC];
      int mtok[TT];
      bool tv[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        n_dg[t].get(dgq[t]);
        n_z[t].get(zq[t]);
        n_xh[t].get(xh[t]);
      }
      if (j0 + TT < g.cols) fetch(j0 + TT);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        tv[t] = rv && (j0 + t < g.cols);
        mtok[t] = tv[t] ? tok_mem<TP>(g, i * g.cols + j0 + t) : 0;
        rs[t] = (tv[t] && p.use_norm) ? p.rstd[(size_t)b * g.L + mtok[t]] : 1.f;
        const float mu = (tv[t] && p.use_norm) ? p.mean[(size_t)b * g.L + mtok[t]] : 0.f;
        if constexpr (TP) {
          if (tv[t] && act) {
            const float* y = yc_r + (size_t)((j0 + t) % tpp) * p.d_in;
#pragma unroll
            for (int v = 0; v < VEC; ++v) ysum[v] = y[v] + y[ydir + v];
          }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) xh[t][v] = (tv[t] && act) ? (0.5f * (ysum[v] + xh[t][v]) - mu) * rs[t] : 0.f;   // skip -> xhat
        float dzv[VEC];
        const float (&dgv)[VEC] = dgq[t];
        const float (&zv)[VEC] = zq[t];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float h = xh[t][v] * lw[v] + lb[v];
          const float sg = fv_sigmoid(zv[v]);
          const float dh = dgv[v] * (zv[v] * sg);
          dzv[v] = dgv[v] * h * (sg * (1.f + zv[v] * (1.f - sg)));
          a_lw[v] = fmaf(dh, xh[t][v], a_lw[v]);
          a_lb[v] += dh;
          dxh[t][v] = dh * lw[v];
          s1 += dxh[t][v];
          s2 = fmaf(dxh[t][v], xh[t][v], s2);
        }
        c1[t] = s1;
        c2[t] = s2;
        if (tv[t] && act) VecIO<T, VEC>::store(dxz_b + (size_t)mtok[t] * 2 * p.d_in + p.d_in + c0, dzv);
      }
      if (p.use_norm) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          c1[t] = wave_sum_uniform(c1[t]);
          c2[t] = wave_sum_uniform(c2[t]);
        }
        if (nch > 1) {      // sum over the row group's channel-chunk waves through LDS
          __syncthreads();
          if (lane == 0)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
              s_red[((rg * 2 + 0) * TT + t) * 16 + cw] = c1[t];
              s_red[((rg * 2 + 1) * TT + t) * 16 + cw] = c2[t];
            }
          __syncthreads();
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            float t1 = 0.f, t2 = 0.f;
            for (int w = 0; w < nch; ++w) {
              t1 += s_red[((rg * 2 + 0) * TT + t) * 16 + w];
              t2 += s_red[((rg * 2 + 1) * TT + t) * 16 + w];
            }
            c1[t] = t1;
            c2[t] = t2;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        float dov[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float d_o = p.use_norm ? rs[t] * (dxh[t][v] - inv_d * (c1[t] + xh[t][v] * c2[t])) : dxh[t][v];
          dov[v] = d_o;
          dyc_acc[v] += 0.5f * d_o;
        }
        if constexpr (TP) {
          if (nopool) {
            if (tv[t] && act) {
              float h[VEC];
#pragma unroll
              for (int v = 0; v < VEC; ++v) h[v] = 0.5f * dov[v];
              VecIO<float, VEC>::store(p.dyc + ((size_t)row * tpp + (j0 + t)) * p.d_in + c0, h);
            }
          } else {
            float* sl = s_dyc + (((j0 + t) % tpp) * nthr + threadIdx.x) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) sl[v] += 0.5f * dov[v];
          }
        }
        if (tv[t] && act) VecIO<T, VEC>::store(dob_b + (size_t)mtok[t] * p.d_in + c0, dov);
      }
    }
    if (rv && act) {
      if constexpr (!TP) {
        VecIO<float, VEC>::store(p.dyc + (size_t)row * p.d_in + c0, dyc_acc);
      } else if (!nopool) {
        for (int c = 0; c < tpp; ++c) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dyc_acc[v] = s_dyc[(c * nthr + threadIdx.x) * VEC + v];
          VecIO<float, VEC>::store(p.dyc + ((size_t)row * tpp + c) * p.d_in + c0, dyc_acc);
        }
      }
    }
  }
  // fixed-order accumulation of the RG row groups into one partial row [d ln_w | d ln_b]
  for (int r = 0; r < RG; ++r) {
    __syncthreads();
    if (r == rg && act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        s_acc[c0 + v] = (r == 0 ? 0.f : s_acc[c0 + v]) + a_lw[v];
        s_acc[p.d_in + c0 + v] = (r == 0 ? 0.f : s_acc[p.d_in + c0 + v]) + a_lb[v];
      }
    }
  }
  __syncthreads();
  float* dst = p.part + (size_t)blockIdx.x * 2 * p.d_in;
  for (int e = threadIdx.x; e < 2 * p.d_in; e += blockDim.x) dst[e] = s_acc[e];
}

// ------------------------------------------------------------------ conv + pool backward (streaming)
// Purely per-channel (no cross-lane traffic): a lane owns VEC channels and streams the row's tokens
// through 4-deep register windows.  Step n consumes token n+3 and produces
//   dpre_f[n+3] (needs x[n..n+3]),  dpre_b[n] (needs x[n..n+3]),  dx[n] (needs dpre_f[n..n+3], dpre_b[n-3..n]).
template <typename T, int VEC, int CH, bool TP, bool PM>   // PM: max pooling (gradient goes to the saved argmax column); CH tokens are fetched (packed, as loaded) ahead of the arithmetic that consumes them
__global__ __launch_bounds__(VEC == 1 ? 1024 : 768) void conv_pool_bwd_kernel(BwdParams p, int nch, int RG) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 12 * d_in accumulator
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform (scalar index math)
  const int rg = wv / nch, cw = wv - rg * nch;
  const int c0 = (cw * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  float Dfh[VEC], Dbh[VEC];
  float a_wf[VEC][CW], a_wb[VEC][CW], a_bf[VEC], a_bb[VEC], a_Df[VEC], a_Db[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    Dfh[v] = act ? 0.5f * p.Df[c0 + v] : 0.f;
    Dbh[v] = act ? 0.5f * p.Db[c0 + v] : 0.f;
    a_bf[v] = a_bb[v] = a_Df[v] = a_Db[v] = 0.f;
#pragma unroll
    for (int k = 0; k < CW; ++k) a_wf[v][k] = a_wb[v][k] = 0.f;
  }
  const int nrows = p.B * g.rows;
  const int tpp = TP ? g.tpp : 1;
  const size_t dstride = (size_t)p.B * g.rows * tpp * p.d_in;
  const int nit = (nrows + gridDim.x * RG - 1) / (gridDim.x * RG);
  for (int it = 0; it < nit; ++it) {
    const int row = (it * gridDim.x + blockIdx.x) * RG + rg;
    if (row < nrows) {          // uniform per wave; no block-level sync inside
      const int b = row / g.rows, i = row - b * g.rows;
      float dcf[3][VEC], dcb[3][VEC];   // pooled gradients of rows i-1, i, i+1 (halo tokens)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ii = i - 1 + r;
        const bool ok = act && !TP && ii >= 0 && ii < g.rows;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const size_t o = ((size_t)b * g.rows + (ok ? ii : 0)) * p.d_in + (act ? c0 + v : 0);
          dcf[r][v] = ok ? p.dxc[o] * p.pool_scale : 0.f;
          dcb[r][v] = ok ? p.dxc[dstride + o] * p.pool_scale : 0.f;
        }
      }
      float af[3][VEC], ab[3][VEC];     // PM: argmax columns of rows i-1, i, i+1
      if constexpr (PM && !TP) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int ii = i - 1 + r;
          const bool ok = act && ii >= 0 && ii < g.rows;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const size_t o = ((size_t)b * g.rows + (ok ? ii : 0)) * p.d_in + (act ? c0 + v : 0);
            af[r][v] = ok ? io<T>::ld((const T*)p.amax + o) : -1.f;
            ab[r][v] = ok ? io<T>::ld((const T*)p.amax + dstride + o) : -1.f;
          }
        }
      }
      const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
      const T* dob_b = (const T*)p.dob_in + (size_t)b * g.L * p.d_in;
      T* dxz_b = (T*)p.dxz + (size_t)b * g.L * 2 * p.d_in;
      const int s_row = i * g.cols;
      // windows: index 0 = token n, 3 = token n+3 (x, do, dpf); dpb: index 0 = token n-3, 3 = token n
      float xw[4][VEC], dw[4][VEC], dpf[4][VEC], dpb[4][VEC];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) xw[k][v] = dw[k][v] = dpf[k][v] = dpb[k][v] = 0.f;
      // preload tokens -3, -2, -1 into slots 1..3 (they shift to 0..2 at the first step)
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const int s = s_row - 4 + k;
        if (s >= 0 && act) {
          const int m = tok_mem<TP>(g, s);
          VecIO<T, VEC>::load(xz_b + (size_t)m * 2 * p.d_in + c0, xw[k]);
          VecIO<T, VEC>::load(dob_b + (size_t)m * p.d_in + c0, dw[k]);
        }
      }
      for (int n0 = -3; n0 < g.cols; n0 += CH) {
        RawVec<T, VEC> xp[CH], dp[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {                    // token n0 + c + 3
          const int j3 = n0 + c + 3, sp = s_row + j3;
          if (act && sp < g.L && j3 < g.cols + 3) {
            const int m = tok_mem<TP>(g, sp);
            xp[c].load(xz_b + (size_t)m * 2 * p.d_in + c0);
            dp[c].load(dob_b + (size_t)m * p.d_in + c0);
          } else {
            xp[c].zero();
            dp[c].zero();
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
        const int n = n0 + c;
        if (n >= g.cols) break;
        // shift the windows by one token and bring in token n+3
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            xw[k][v] = xw[k + 1][v];
            dw[k][v] = dw[k + 1][v];
            dpf[k][v] = dpf[k + 1][v];
            dpb[k][v] = dpb[k + 1][v];
          }
        const int s3 = s_row + n + 3;
        const bool v3 = s3 < g.L;                         // token n+3 exists (s3 >= 0 always here)
        const bool v0 = s_row + n >= 0;                   // token n exists
        xp[c].get(xw[3]);
        dp[c].get(dw[3]);
        const int r3 = (n + 3 >= g.cols) ? 2 : 1;         // row of token n+3 relative to i-1
        const int r0 = (n < 0) ? 0 : 1;                   // row of token n
        const bool own3 = n + 3 < g.cols;                 // token n+3 belongs to this row (n+3 >= 0 always)
        const bool own0 = n >= 0;
        float cf_t[VEC], cb_t[VEC];                       // channel-wise tokenization: pooled gradient of (row, channel slot)
#pragma unroll
        for (int v = 0; v < VEC; ++v) cf_t[v] = cb_t[v] = 0.f;
        float af_t[VEC], ab_t[VEC];                       // PM: argmax column of that (row, channel slot)
#pragma unroll
        for (int v = 0; v < VEC; ++v) af_t[v] = ab_t[v] = -1.f;
        const int jf3 = n + 3 >= g.cols ? n + 3 - g.cols : n + 3;      // position of token n+3 inside its own row
        const int jf0 = n < 0 ? n + g.cols : n;
        if (TP && act) {
          const int i3 = i - 1 + r3, i0 = i - 1 + r0;
          const int sl3 = (n + 3 + tpp * 4) % tpp, sl0 = (n + tpp * 4) % tpp;     // n >= -3 > -4*tpp
          if (i3 >= 0 && i3 < g.rows) {
            const size_t o = (((size_t)b * g.rows + i3) * tpp + sl3) * p.d_in + c0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              cf_t[v] = p.dxc[o + v] * p.pool_scale;
              if constexpr (PM) af_t[v] = io<T>::ld((const T*)p.amax + o + v);
            }
          }
          if (i0 >= 0 && i0 < g.rows) {
            const size_t o = (((size_t)b * g.rows + i0) * tpp + sl0) * p.d_in + c0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              cb_t[v] = p.dxc[dstride + o + v] * p.pool_scale;
              if constexpr (PM) ab_t[v] = io<T>::ld((const T*)p.amax + dstride + o + v);
            }
          }
        }
        const float col3 = (float)(TP ? jf3 / tpp : jf3), col0 = (float)(TP ? jf0 / tpp : jf0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float pf = cp.bf[v], pb = cp.bb[v];
#pragma unroll
          for (int k = 0; k < CW; ++k) {
            pf = fmaf(cp.wf[v][k], xw[k][v], pf);         // pre_f[n+3] = b + sum_k w[k] x[n+k]
            pb = fmaf(cp.wb[v][k], xw[3 - k][v], pb);     // pre_b[n]   = b + sum_k w[k] x[n+3-k]
          }
          const float sgf = fv_sigmoid(pf), sgb = fv_sigmoid(pb);
          const float dsf = sgf * (1.f + pf * (1.f - sgf)), dsb = sgb * (1.f + pb * (1.f - sgb));
          float cf = TP ? cf_t[v] : (r3 == 2 ? dcf[2][v] : dcf[1][v]);
          float cb = TP ? cb_t[v] : (r0 == 0 ? dcb[0][v] : dcb[1][v]);
          if constexpr (PM) {      // max pooling: only the argmax token of a pooling group receives the gradient
            const float a3 = TP ? af_t[v] : (r3 == 2 ? af[2][v] : af[1][v]);
            const float a0 = TP ? ab_t[v] : (r0 == 0 ? ab[0][v] : ab[1][v]);
            cf = a3 == col3 ? cf : 0.f;
            cb = a0 == col0 ? cb : 0.f;
          }
          const float nf = v3 ? (Dfh[v] * dw[3][v] + cf) * dsf : 0.f;
          const float nb = v0 ? (Dbh[v] * dw[0][v] + cb) * dsb : 0.f;
          dpf[3][v] = nf;
          dpb[3][v] = nb;
          if (own3) {
#pragma unroll
            for (int k = 0; k < CW; ++k) a_wf[v][k] = fmaf(nf, xw[k][v], a_wf[v][k]);
            a_bf[v] += nf;
            a_Df[v] = fmaf(0.5f * dw[3][v], pf * sgf, a_Df[v]);
          }
          if (own0) {
#pragma unroll
            for (int k = 0; k < CW; ++k) a_wb[v][k] = fmaf(nb, xw[3 - k][v], a_wb[v][k]);
            a_bb[v] += nb;
            a_Db[v] = fmaf(0.5f * dw[0][v], pb * sgb, a_Db[v]);
          }
        }
        if (own0) {
          float dx[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < CW; ++k) {
              acc = fmaf(cp.wf[v][k], dpf[3 - k][v], acc);   // dpre_f[n+3-k]
              acc = fmaf(cp.wb[v][k], dpb[k][v], acc);       // dpre_b[n-3+k]
            }
            dx[v] = acc;
          }
          if (act) VecIO<T, VEC>::store(dxz_b + (size_t)tok_mem<TP>(g, s_row + n) * 2 * p.d_in + c0, dx);
        }
        }
      }
    }
  }
  // one partial row per block: [d w (d_in*4) | d w_b (d_in*4) | d b | d b_b | dD | dD_b]
  const int D = p.d_in;
  for (int r = 0; r < RG; ++r) {
    __syncthreads();
    if (r == rg && act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int c = c0 + v;
#pragma unroll
        for (int k = 0; k < CW; ++k) {
          smem[c * 4 + k] = (r == 0 ? 0.f : smem[c * 4 + k]) + a_wf[v][k];
          smem[4 * D + c * 4 + k] = (r == 0 ? 0.f : smem[4 * D + c * 4 + k]) + a_wb[v][k];
        }
        smem[8 * D + c] = (r == 0 ? 0.f : smem[8 * D + c]) + a_bf[v];
        smem[9 * D + c] = (r == 0 ? 0.f : smem[9 * D + c]) + a_bb[v];
        smem[10 * D + c] = (r == 0 ? 0.f : smem[10 * D + c]) + a_Df[v];
        smem[11 * D + c] = (r == 0 ? 0.f : smem[11 * D + c]) + a_Db[v];
      }
    }
  }
  __syncthreads();
  float* dst = p.part + (size_t)blockIdx.x * 12 * D;
  for (int e = threadIdx.x; e < 12 * D; e += blockDim.x) dst[e] = smem[e];
}

// out[i] = sum_s in[s*n + i] in a fixed order.  Block = 32 outputs x 8 row-slices: slice q sums rows
// q, q+8, ... (independent 128-B coalesced loads), then the 8 slice sums are added in order.
// Sum of rows q, q+8, q+16, ... of column i.  Eight independent loads are in flight per thread: the reduction is
// bound by memory-level parallelism (a block column is only 128 B wide), not by bandwidth per request.
__device__ __forceinline__ float column_sum(const float* __restrict__ in, int S, size_t n, size_t i, int q) {
  if (S >= 128) {
    // many partials (one per block of a persistent backward kernel: 256): sixteen loads in flight per thread, the
    // block's life is round trips to memory and little else
    float b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) b[u] = 0.f;
    int s = q;
    for (; s + 120 < S; s += 128) {
#pragma unroll
      for (int u = 0; u < 16; ++u) b[u] += in[(size_t)(s + 8 * u) * n + i];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (s + 8 * u < S) b[u] += in[(size_t)(s + 8 * u) * n + i];
    return (((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]))) +
           (((b[8] + b[9]) + (b[10] + b[11])) + ((b[12] + b[13]) + (b[14] + b[15])));
  }
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  int s = q;
  for (; s + 56 < S; s += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += in[(size_t)(s + 8 * u) * n + i];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (s + 8 * u < S) a[u] += in[(size_t)(s + 8 * u) * n + i];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

constexpr int FLAT_S = 8;      // at most this many partials: one column per thread (see flat_sum)

// Few partials (split-K weight gradients: 7): one column per thread, every partial and the old value in flight at once,
// no LDS stage.  A 32-column block would move 32 x 7 floats behind two dependent memory round trips; at FastVim-T
// that was 31 872 such blocks per launch, 37 us for 26 MB.
__device__ __forceinline__ void flat_sum(const float* __restrict__ in, float* __restrict__ out, int S, size_t n, size_t i,
                                         int accumulate) {
  if (i >= n) return;
  float v[FLAT_S];
#pragma unroll
  for (int u = 0; u < FLAT_S; ++u) v[u] = u < S ? in[(size_t)u * n + i] : 0.f;
  const float o = accumulate ? out[i] : 0.f;
  out[i] = o + (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int S, size_t n, int accumulate) {
  __shared__ float s_acc[8][33];
  if (S <= FLAT_S) {
    flat_sum(in, out, S, n, (size_t)blockIdx.x * 256 + threadIdx.x, accumulate);
    return;
  }
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + c;
  s_acc[q][c] = i < n ? column_sum(in, S, n, i, q) : 0.f;
  __syncthreads();
  if (q == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_acc[k][c];
    out[i] = accumulate ? out[i] + t : t;
  }
}

int rg_combine(int d_in, int VEC) { int nch = fv_cdiv(d_in, 64 * VEC); int r = 8 / nch; return r < 1 ? 1 : (r > RGMAX ? RGMAX : r); }
int rg_convpool(int d_in, int VEC) { int nch = fv_cdiv(d_in, 64 * VEC); int r = (VEC == 1 ? 16 : 12) / nch; return r < 1 ? 1 : (r > RGMAX ? RGMAX : r); }
int vec_combine(int d_in, int tpp) {
  if (tpp > 1)      // LDS slot accumulators: keep the per-lane state small
    return (d_in % 128 == 0 && d_in <= 8 * 128) ? 2 : (d_in % 256 == 0 && d_in <= 8 * 256) ? 4 : 1;
  return (d_in % 384 == 0 && d_in <= 8 * 384) ? 6 : (d_in % 256 == 0 && d_in <= 8 * 256) ? 4 : 1;
}
int vec_convpool(int d_in) { return (d_in % 128 == 0 && d_in <= 12 * 128) ? 2 : 1; }
int persistent_blocks(long nrows, int rg) {
  long groups = (nrows + rg - 1) / rg;
  // at most one block per CU: measured best for the persistent backward kernels (conv_pool_bwd at FastVim-T 32.6 us with 256
  // blocks vs 37.8 with 448, 36.9 with 384, 46.0 with 299) -- no CU carries two blocks while others carry one, and the
  // per-block gradient partials halve.  Below that cap the grid is the one that deals the row groups out EVENLY: 448 groups
  // (FastVim-T: 1 792 pooling rows, 4 per block and iteration) are 2 iterations of 224 blocks, not 1.75 of 256 (round 5,
  // same box, step 5.39 -> 5.36 ms; in round 2 the same 224 measured 33.3 against 32.6 us)
  static const int cap = fv_tune("FASTVIM_BWD_GRID", 256);   // tuning hook
  const long per = (groups + cap - 1) / cap;
  return (int)((groups + per - 1) / per);
}

// Grid of the conv + pool adjoint.  Channel-wise tokenization (tokens_per_patch 8) runs the cell-walking kernel, whose blocks
// are four waves -- `groups` of them per row over blockIdx.y, two resident per CU (convpool_bwd_row.hip): the launch costs
// (rounds of resident blocks) x (iterations per block), so a grid that keeps all blocks resident in ONE round can beat one
// block per CU in x: FastChannelVim-S (896 rows, 3 groups) 224 x 3 blocks x 2 iterations = 2 rounds x 2 -> 150 x 3 x 3 =
// 1 round x 3; 177 -> 166 us (profiles/r05_hook_sweep_final_tree.log).  Everything else: persistent_blocks.
int conv_pool_bwd_blocks(long nrows, int d_in, int tpp) {
  const int VEC = vec_convpool(d_in), rg = rg_convpool(d_in, VEC);
  const int base = persistent_blocks(nrows, rg);
  if (tpp != 8 || VEC != 2 || d_in % 128) return base;
  const int nch = d_in / 128;
  int groups = (nch + 1) / 2;
  while (nch % groups) ++groups;
  const int rgr = 4 / (nch / groups);
  const long rgroups = (nrows + rgr - 1) / rgr, slots = 2L * fv_cu_count();
  auto even = [&](long cap) { const long per = (rgroups + cap - 1) / cap; return (rgroups + per - 1) / per; };
  auto cost = [&](long x) { return ((x * groups + slots - 1) / slots) * ((rgroups + x - 1) / x); };
  const long one_round = even(slots / groups > 0 ? slots / groups : 1);
  return cost(one_round) < cost(base) ? (int)one_round : base;
}

template <typename T, int VEC>
int launch_combine_bwd(const BwdParams& p, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 8), "mixer_combine_bwd: d_inner %d too large for the VEC=%d row walker", p.d_in, VEC);
  const int rg = rg_combine(p.d_in, VEC);
  dim3 grid(persistent_blocks((long)p.B * p.geo.rows, rg)), block(64 * nch * rg);
  const size_t extra = (p.geo.tpp > 1 && p.geo.pcols > 1) ? (size_t)p.geo.tpp * 64 * nch * rg * VEC : 0;
  FV_CHECK((RGMAX * 64 + 2 * p.d_in + extra) * 4 <= 64 * 1024, "mixer_combine_bwd: tokens_per_patch %d too large", p.geo.tpp);
  const bool tp = p.geo.tpp > 1;
  if (p.geo.cols % 2 == 0) {
    size_t smem = (size_t)(RGMAX * 2 * 2 * 16 + 2 * p.d_in + extra) * 4;
    if (tp) hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 2, true>), grid, block, smem, st, p, nch, rg);
    else hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 2, false>), grid, block, smem, st, p, nch, rg);
  } else {
    size_t smem = (size_t)(RGMAX * 2 * 1 * 16 + 2 * p.d_in + extra) * 4;
    if (tp) hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 1, true>), grid, block, smem, st, p, nch, rg);
    else hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 1, false>), grid, block, smem, st, p, nch, rg);
  }
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T, int VEC>
int launch_conv_pool_bwd(const BwdParams& p, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 12), "mixer_conv_pool_bwd: d_inner %d too large for the VEC=%d row walker", p.d_in, VEC);
  const int rg = rg_convpool(p.d_in, VEC);
  dim3 grid(conv_pool_bwd_blocks((long)p.B * p.geo.rows, p.d_in, p.geo.tpp)), block(64 * nch * rg);
  size_t smem = (size_t)12 * p.d_in * 4;
  FV_CHECK(smem <= 160 * 1024, "mixer_conv_pool_bwd: d_inner %d too large", p.d_in);
  if (smem > 64 * 1024) {     // opt in to > 64 KiB of dynamic LDS (once per instantiation; not a stream operation)
    static FvOncePerDevice done;   
    if (done.first()) {
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_kernel<T, VEC, 17, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_kernel<T, VEC, 8, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_kernel<T, VEC, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)0;     
    }
  }
  // short rows: the whole-row kernel (convpool_bwd_row.hip) over the same persistent grid / partial layout
  static const bool rowk = (fv_tune("FASTVIM_BWD_ROWK", 1) != 0);   // tuning hook
  if (rowk && VEC == 2 && !p.amax) {
    int rc = fvi::conv_pool_bwd_row(p, nch, rg, (int)grid.x, smem, sizeof(T) == 4 ? FV_F32 : FV_BF16, st);
    if (rc != FV_ERR_UNSUPPORTED) return rc;
  }
  FV_CHECK(!p.dxc2, "mixer_conv_pool_bwd2: a second pooled-gradient addend is taken by the whole-row kernel only "
                    "(mean pooling, tokens_per_patch 1, 14 or 16 columns, d_inner a multiple of 128)");
  if (p.amax) {      // max pooling: generic kernels only
    if (smem > 64 * 1024) {
      static FvOncePerDevice done_pm;   
      if (done_pm.first()) {
        (void)hipFuncSetAttribute((const void*)conv_pool_bwd_kernel<T, VEC, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_pool_bwd_kernel<T, VEC, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)0;        
      }
    }
    if (p.geo.tpp > 1) hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 8, true, true>), grid, block, smem, st, p, nch, rg);
    else hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 8, false, true>), grid, block, smem, st, p, nch, rg);
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  if (p.geo.tpp > 1) hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 8, true, false>), grid, block, smem, st, p, nch, rg);
  else if (p.geo.cols + 3 <= 17) hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 17, false, false>), grid, block, smem, st, p, nch, rg);   // whole row in flight
  else hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 8, false, false>), grid, block, smem, st, p, nch, rg);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T>
int dispatch_bwd(int which, const BwdParams& p, hipStream_t st) {
  if (which == 0) {
    {
      int rc = fvi::combine_bwd_wave(p, sizeof(T) == 4 ? FV_F32 : FV_BF16, st);
      if (rc != FV_ERR_UNSUPPORTED) return rc;
    }
    const int v = vec_combine(p.d_in, p.geo.tpp);
    if (v == 6) return launch_combine_bwd<T, 6>(p, st);
    if (v == 4) return launch_combine_bwd<T, 4>(p, st);
    if (v == 2) return launch_combine_bwd<T, 2>(p, st);
    return launch_combine_bwd<T, 1>(p, st);
  }
  if (vec_convpool(p.d_in) == 2) return launch_conv_pool_bwd<T, 2>(p, st);
  return launch_conv_pool_bwd<T, 1>(p, st);
}

int check_geo_b(int B, int rows, int cols, int s_i, int s_j, int d_in, int dtype) {
  FV_CHECK(B > 0 && rows > 0 && cols > 0 && d_in > 0, "mixer: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer: dtype must be fp32 or bf16");
  FV_CHECK((s_i == cols && s_j == 1) || (s_i == 1 && s_j == rows),
           "mixer: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid", s_i, s_j, rows, cols);
  return FV_OK;
}

}  // namespace

extern "C" int fv_mixer_bwd_blocks(int batch, int rows, int d_inner, int tokens_per_patch, int which) {
  const long n = (long)batch * rows;
  if (which == 0) {
    const int wb = fvi::combine_wave_blocks(batch, rows, tokens_per_patch, d_inner);
    if (wb) return wb;
  }
  return which == 0 ? persistent_blocks(n, rg_combine(d_inner, vec_combine(d_inner, tokens_per_patch)))
                    : conv_pool_bwd_blocks(n, d_inner, tokens_per_patch);
}

extern "C" int fv_mixer_combine_bwd(const void* dg, const void* xz, const void* skip, const float* yc,
                                    const float* ln_w, const float* ln_b, const float* mean, const float* rstd,
                                    void* dxz, void* d_o, float* dyc,
                                    float* partials, int batch, int rows, int cols, int tok_stride_row,
                                    int tok_stride_col, int tokens_per_patch, int d_inner, int dtype,
                                    fv_stream_t stream) {
  int rc = check_geo_b(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype);
  FV_CHECK(tokens_per_patch > 0, "mixer_combine_bwd: tokens_per_patch must be positive");
  if (rc) return rc;
  FV_CHECK(dg && xz && skip && yc && dxz && d_o && dyc && partials, "mixer_combine_bwd: null pointer");
  FV_CHECK(!ln_w || (ln_b && mean && rstd), "mixer_combine_bwd: LayerNorm needs weight, bias, mean, rstd");
  BwdParams p{};
  p.dg = dg; p.xz = xz; p.skip = skip; p.yc = yc; p.lnw = ln_w; p.lnb = ln_b; p.mean = mean; p.rstd = rstd;
  p.dxz = dxz; p.dob = d_o; p.dyc = dyc; p.part = partials;
  p.use_norm = ln_w != nullptr;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner;
  return dtype == FV_F32 ? dispatch_bwd<float>(0, p, (hipStream_t)stream)
                         : dispatch_bwd<bf16_t>(0, p, (hipStream_t)stream);
}

extern "C" int fv_mixer_conv_pool_bwd(const void* xz, const void* d_o, const float* dxc, const float* conv_w,
                                      const float* conv_b, const float* conv_w_b, const float* conv_b_b,
                                      const float* D, const float* D_b, const void* amax, void* dxz,
                                      float* partials, int batch,
                                      int rows, int cols, int tok_stride_row, int tok_stride_col,
                                      int tokens_per_patch, int d_inner, int d_conv, int pool_max,
                                      float scaling_factor, int dtype, fv_stream_t stream) {
  return fv_mixer_conv_pool_bwd2(xz, d_o, dxc, nullptr, conv_w, conv_b, conv_w_b, conv_b_b, D, D_b, amax, dxz, partials, batch,
                                 rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch, d_inner, d_conv, pool_max,
                                 scaling_factor, dtype, stream);
}

extern "C" int fv_mixer_conv_pool_bwd2_ok(int rows, int cols, int tokens_per_patch, int d_inner, int pool_max) {
  return !pool_max && tokens_per_patch == 1 && (cols == 14 || cols == 16) && rows > 0 && d_inner % 128 == 0 && d_inner <= 2048;
}

extern "C" int fv_mixer_conv_pool_bwd2(const void* xz, const void* d_o, const float* dxc, const void* dxc2,
                                       const float* conv_w, const float* conv_b, const float* conv_w_b,
                                       const float* conv_b_b, const float* D, const float* D_b, const void* amax,
                                       void* dxz, float* partials, int batch, int rows, int cols, int tok_stride_row,
                                       int tok_stride_col, int tokens_per_patch, int d_inner, int d_conv, int pool_max,
                                       float scaling_factor, int dtype, fv_stream_t stream) {
  int rc = check_geo_b(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype);
  FV_CHECK(tokens_per_patch > 0, "mixer_conv_pool_bwd: tokens_per_patch must be positive");
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  FV_CHECK(!pool_max || amax, "mixer_conv_pool_bwd: max pooling needs the argmax columns saved by the forward");
  FV_CHECK(cols * tokens_per_patch >= 3, "mixer_conv_pool_bwd: needs at least 3 tokens per pooling row (got %d)", cols * tokens_per_patch);
  FV_CHECK(xz && d_o && dxc && conv_w && conv_w_b && D && D_b && dxz && partials, "mixer_conv_pool_bwd: null pointer");
  BwdParams p{};
  p.xz = xz; p.dob_in = d_o; p.dxc = dxc; p.dxc2 = dxc2; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b;
  p.Df = D; p.Db = D_b; p.dxz = dxz; p.part = partials;
  FV_CHECK(!dxc2 || fv_mixer_conv_pool_bwd2_ok(rows, cols, tokens_per_patch, d_inner, pool_max),
           "mixer_conv_pool_bwd2: shape does not take a second pooled-gradient addend (fv_mixer_conv_pool_bwd2_ok)");
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner;
  p.pool_scale = pool_max ? 1.f : scaling_factor / (float)cols;
  p.amax = pool_max ? amax : nullptr;
  return dtype == FV_F32 ? dispatch_bwd<float>(1, p, (hipStream_t)stream)
                         : dispatch_bwd<bf16_t>(1, p, (hipStream_t)stream);
}

// Several independent fixed-order reductions in ONE launch (gradient partials of different kernels /
// layers whose sums are only needed before the optimizer step).
// The job table rides in the kernel arguments: 4 KiB less 256 bytes of implicit arguments.  Round 6 packs a job into 18
// bytes -- both pointers as 32-bit FLOAT offsets from the lowest pointer of the launch (16 GB of reach; a launch whose
// buffers lie further apart is cut in two), 32-bit lengths, 16-bit partial counts -- 208 jobs per launch instead of 96: a
// FastVim-T step's 197 jobs are ONE launch (three before), and the big and small jobs of a launch fill each other's gaps.
constexpr int MAXJOBS = 208;
struct ReduceJobs {
  const float* in_base;
  float* out_base;
  unsigned in_off[MAXJOBS];      // partials = in_base + in_off (floats)
  unsigned out_off[MAXJOBS];     // out = out_base + out_off (floats)
  unsigned n[MAXJOBS];           // elements per partial (< 2^32: checked at the launch)
  int blk_end[MAXJOBS];          // exclusive prefix of blocks per job
  unsigned short S[MAXJOBS];     // partials (<= 65 535: checked at the launch)
  int njobs, accumulate;
};
static_assert(sizeof(ReduceJobs) <= 4096 - 256, "job table must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceJobs J) {
  __shared__ float s_acc[8][33];
  int lo = 0, hi = J.njobs - 1;         // first job whose block range ends beyond this block (binary search, uniform)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= J.blk_end[mid]) lo = mid + 1; else hi = mid;
  }
  const int job = lo;
  const int blk = blockIdx.x - (job ? J.blk_end[job - 1] : 0);
  const float* __restrict__ in = J.in_base + J.in_off[job];
  float* __restrict__ out = J.out_base + J.out_off[job];
  const int S = J.S[job];
  const size_t n = (size_t)J.n[job];
  if (S <= FLAT_S) {
    flat_sum(in, out, S, n, (size_t)blk * 256 + threadIdx.x, J.accumulate);
    return;
  }
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t i = (size_t)blk * 32 + c;
  s_acc[q][c] = i < n ? column_sum(in, S, n, i, q) : 0.f;
  __syncthreads();
  if (q == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_acc[k][c];
    out[i] = J.accumulate ? out[i] + t : t;
  }
}

static int reduce_multi_launch(const float* const* partials, float* const* outs, const int* n_partials, const size_t* ns,
                               int njobs, int accumulate, hipStream_t st) {
  uintptr_t in_lo = ~(uintptr_t)0, in_hi = 0, out_lo = ~(uintptr_t)0, out_hi = 0;
  for (int j = 0; j < njobs; ++j) {
    const uintptr_t a = (uintptr_t)partials[j], b = (uintptr_t)outs[j];
    in_lo = a < in_lo ? a : in_lo; in_hi = a > in_hi ? a : in_hi;
    out_lo = b < out_lo ? b : out_lo; out_hi = b > out_hi ? b : out_hi;
  }
  const uintptr_t reach = (uintptr_t)0xffffffffull * 4;
  if (njobs > 1 && (in_hi - in_lo > reach || out_hi - out_lo > reach)) {      // buffers too far apart for 32-bit offsets
    const int h = njobs / 2;
    int rc = reduce_multi_launch(partials, outs, n_partials, ns, h, accumulate, st);
    if (rc) return rc;
    return reduce_multi_launch(partials + h, outs + h, n_partials + h, ns + h, njobs - h, accumulate, st);
  }
  ReduceJobs J{};
  J.in_base = (const float*)in_lo; J.out_base = (float*)out_lo;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    J.in_off[j] = (unsigned)(((uintptr_t)partials[j] - in_lo) >> 2);
    J.out_off[j] = (unsigned)(((uintptr_t)outs[j] - out_lo) >> 2);
    J.S[j] = (unsigned short)n_partials[j]; J.n[j] = (unsigned)ns[j];
    blocks += fv_cdiv((long)ns[j], n_partials[j] <= FLAT_S ? 256 : 32);
    J.blk_end[j] = blocks;
  }
  J.njobs = njobs; J.accumulate = accumulate;
  if (blocks == 0) return FV_OK;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(256), 0, st, J);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_reduce_partials_multi(const float* const* partials, float* const* outs, const int* n_partials,
                                        const size_t* ns, int njobs, int accumulate, fv_stream_t stream) {
  FV_CHECK(partials && outs && n_partials && ns && njobs > 0 && njobs <= MAXJOBS,
           "reduce_partials_multi: 1..%d jobs", MAXJOBS);
  for (int j = 0; j < njobs; ++j) {
    FV_CHECK(partials[j] && outs[j] && n_partials[j] > 0, "reduce_partials_multi: bad job %d", j);
    FV_CHECK(n_partials[j] <= 65535 && ns[j] < 0xffffffffull, "reduce_partials_multi: job %d too large for the packed table", j);
    FV_CHECK((((uintptr_t)partials[j] | (uintptr_t)outs[j]) & 3) == 0, "reduce_partials_multi: job %d is not float-aligned", j);
  }
  return reduce_multi_launch(partials, outs, n_partials, ns, njobs, accumulate, (hipStream_t)stream);
}

extern "C" int fv_reduce_partials(const float* partials, float* out, int n_partials, size_t n, int accumulate,
                                  fv_stream_t stream) {
  FV_CHECK(partials && out && n_partials > 0, "reduce_partials: bad arguments");
  if (n == 0) return FV_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(fv_cdiv((long)n, n_partials <= FLAT_S ? 256 : 32)), dim3(256), 0, (hipStream_t)stream,
                     partials, out, n_partials, n, accumulate);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
