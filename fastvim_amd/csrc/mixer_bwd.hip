// Backward kernels of the fused FastVim mixer "middle" (channel-last).  Hand-written adjoint of
// csrc/mixer_fwd.hip; replaces the autograd graph of mamba_simple_faster.py:270-444 and the
// hand-written backward of FastVim_MambaInnerFnNoOutProj_withoutZ
// (mamba_ssm/ops/selective_scan_interface.py:607-776):
//
//   fv_mixer_combine_bwd   : d(gate), d(LayerNorm), d(average) -> dz, do, per-row pooled
//                            dyc = 0.5*sum_j do, partials of dLN.weight/bias, dD, dD_b.
//   fv_mixer_scan_bwd      : adjoint recurrence of the pooled scan for both directions with the
//                            dt_proj adjoint fused in: writes d(x_dbl) = [d dt_low | dB | dC]
//                            (reduced over channels in-kernel, deterministically), d(xc) through
//                            the scan, and per-batch partials of dA_log, d dt_proj.weight/bias.
//   fv_mixer_conv_pool_bwd : adjoint of mean-pool + SiLU + both depthwise convs -> dx, and
//                            partials of the conv weight/bias gradients.
//   fv_reduce_partials     : fixed-order sum of per-block partials (no float atomics anywhere).
#include "mixer_common.h"

namespace {

struct BwdParams {
  const void *xz, *dg, *dob_in;
  const float *yc, *wf, *bf, *wb, *bb, *Df, *Db, *lnw, *lnb, *mean, *rstd, *dxc;
  void *dxz, *dob;
  float *dyc, *part;
  Geo geo;
  int B, d_in, use_norm;
  float pool_scale;
};

template <int VEC, int TJ>
__device__ __forceinline__ void conv_pre(const ChanParams<VEC>& cp, const float (&x)[TJ + 6][VEC], int k,
                                         float (&pf)[VEC], float (&pb)[VEC], bool want_f, bool want_b) {
  // pre-activations at tile index k (token j0-3+k): forward needs x[k-3..k], backward x[k..k+3]
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float f = cp.bf[v], bk = cp.bb[v];
#pragma unroll
    for (int kk = 0; kk < CW; ++kk) {
      const int kf = k - 3 + kk < 0 ? 0 : k - 3 + kk;            // clamped: unused when !want_f
      const int kb = k + 3 - kk > TJ + 5 ? TJ + 5 : k + 3 - kk;  // clamped: unused when !want_b
      if (want_f) f = fmaf(cp.wf[v][kk], x[kf][v], f);
      if (want_b) bk = fmaf(cp.wb[v][kk], x[kb][v], bk);
    }
    pf[v] = f;
    pb[v] = bk;
  }
}

// cross-wave sum of TJ wave-uniform values through LDS (block = nw waves of one row)
template <int TJ>
__device__ __forceinline__ void block_sum(float (&s)[TJ], float* s_red, int wv, int nw, int lane) {
  if (nw == 1) return;
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) s_red[wv * TJ + jj] = s[jj];
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < TJ; ++jj) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += s_red[w * TJ + jj];
    s[jj] = t;
  }
}

// ------------------------------------------------------------------ combine backward
template <typename T, int VEC, int TJ>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void combine_bwd_kernel(BwdParams p) {
  __shared__ float s_red[16 * TJ];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c0 = (wv * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  float Df[VEC], Db[VEC], lw[VEC], lb[VEC];
  float a_lw[VEC], a_lb[VEC], a_Df[VEC], a_Db[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    Df[v] = act ? p.Df[c0 + v] : 0.f;
    Db[v] = act ? p.Db[c0 + v] : 0.f;
    lw[v] = act && p.use_norm ? p.lnw[c0 + v] : 1.f;
    lb[v] = act && p.use_norm ? p.lnb[c0 + v] : 0.f;
    a_lw[v] = a_lb[v] = a_Df[v] = a_Db[v] = 0.f;
  }
  const float inv_d = 1.f / (float)p.d_in;
  const int nrows = p.B * g.rows;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = row / g.rows, i = row - b * g.rows;
    float ysum[VEC], dyc_acc[VEC];
    {
      size_t o = (size_t)row * p.d_in + c0;
      size_t dstride = (size_t)p.B * g.rows * p.d_in;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        ysum[v] = act ? p.yc[o + v] + p.yc[dstride + o + v] : 0.f;
        dyc_acc[v] = 0.f;
      }
    }
    const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
    const T* dg_b = (const T*)p.dg + (size_t)b * g.L * p.d_in;
    T* dxz_b = (T*)p.dxz + (size_t)b * g.L * 2 * p.d_in;
    T* dob_b = (T*)p.dob + (size_t)b * g.L * p.d_in;
    for (int j0 = 0; j0 < g.cols; j0 += TJ) {
      float x[TJ + 6][VEC];
      load_x_tile<T, VEC, TJ, 3>(xz_b, g, p.d_in, i, j0, c0, act, x);
      float xf[TJ][VEC], xb[TJ][VEC], xh[TJ][VEC], dxh[TJ][VEC], c1[TJ], c2[TJ], rs[TJ];
      int mtok[TJ];
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) {
        const bool valid = j0 + jj < g.cols;
        float pf[VEC], pb[VEC];
        conv_pre<VEC, TJ>(cp, x, jj + 3, pf, pb, true, true);
        mtok[jj] = valid ? tok_mem(g, i * g.cols + j0 + jj) : 0;
        float mu = 0.f;
        rs[jj] = 1.f;
        if (p.use_norm && valid) {
          mu = p.mean[(size_t)b * g.L + mtok[jj]];
          rs[jj] = p.rstd[(size_t)b * g.L + mtok[jj]];
        }
        float dgv[VEC], zv[VEC], dzv[VEC];
        if (valid && act) {
          VecIO<T, VEC>::load(dg_b + (size_t)mtok[jj] * p.d_in + c0, dgv);
          VecIO<T, VEC>::load(xz_b + (size_t)mtok[jj] * 2 * p.d_in + p.d_in + c0, zv);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dgv[v] = zv[v] = 0.f;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          xf[jj][v] = fv_silu(pf[v]);
          xb[jj][v] = fv_silu(pb[v]);
          float o = 0.5f * (ysum[v] + Df[v] * xf[jj][v] + Db[v] * xb[jj][v]);
          xh[jj][v] = (o - mu) * rs[jj];
          float h = xh[jj][v] * lw[v] + lb[v];
          float sg = fv_sigmoid(zv[v]);
          float sz = zv[v] * sg;
          float dh = dgv[v] * sz;
          dzv[v] = dgv[v] * h * (sg * (1.f + zv[v] * (1.f - sg)));
          a_lw[v] += dh * xh[jj][v];
          a_lb[v] += dh;
          dxh[jj][v] = dh * lw[v];
          s1 += dxh[jj][v];
          s2 += dxh[jj][v] * xh[jj][v];
        }
        c1[jj] = s1;
        c2[jj] = s2;
        if (valid && act) VecIO<T, VEC>::store(dxz_b + (size_t)mtok[jj] * 2 * p.d_in + p.d_in + c0, dzv);
      }
      if (p.use_norm) {
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
          c1[jj] = wave_sum_uniform(c1[jj]);
          c2[jj] = wave_sum_uniform(c2[jj]);
        }
        block_sum<TJ>(c1, s_red, wv, nw, lane);
        block_sum<TJ>(c2, s_red, wv, nw, lane);
      }
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) {
        const bool valid = j0 + jj < g.cols;
        float dov[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float d_o = p.use_norm ? rs[jj] * (dxh[jj][v] - inv_d * (c1[jj] + xh[jj][v] * c2[jj])) : dxh[jj][v];
          if (!valid || !act) d_o = 0.f;
          dov[v] = d_o;
          a_Df[v] += 0.5f * d_o * xf[jj][v];
          a_Db[v] += 0.5f * d_o * xb[jj][v];
          dyc_acc[v] += 0.5f * d_o;
        }
        if (valid && act) VecIO<T, VEC>::store(dob_b + (size_t)mtok[jj] * p.d_in + c0, dov);
      }
    }
    if (act) VecIO<float, VEC>::store(p.dyc + (size_t)row * p.d_in + c0, dyc_acc);
  }
  if (act) {
    float* dst = p.part + ((size_t)blockIdx.x * p.d_in + c0) * 4;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      dst[v * 4 + 0] = a_lw[v];
      dst[v * 4 + 1] = a_lb[v];
      dst[v * 4 + 2] = a_Df[v];
      dst[v * 4 + 3] = a_Db[v];
    }
  }
}

// ------------------------------------------------------------------ conv + pool backward
template <typename T, int VEC, int TJ>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void conv_pool_bwd_kernel(BwdParams p) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c0 = (wv * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  float Dfh[VEC], Dbh[VEC];
  float a_wf[VEC][CW], a_wb[VEC][CW], a_bf[VEC], a_bb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    Dfh[v] = act ? 0.5f * p.Df[c0 + v] : 0.f;
    Dbh[v] = act ? 0.5f * p.Db[c0 + v] : 0.f;
    a_bf[v] = a_bb[v] = 0.f;
#pragma unroll
    for (int k = 0; k < CW; ++k) a_wf[v][k] = a_wb[v][k] = 0.f;
  }
  const int nrows = p.B * g.rows;
  const size_t dstride = (size_t)p.B * g.rows * p.d_in;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = row / g.rows, i = row - b * g.rows;
    // pooled-gradient of rows i-1, i, i+1 (halo tokens belong to the neighbouring rows)
    float dcf[3][VEC], dcb[3][VEC];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int ii = i - 1 + r;
      bool ok = act && ii >= 0 && ii < g.rows;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        size_t o = ((size_t)b * g.rows + ii) * p.d_in + c0 + v;
        dcf[r][v] = ok ? p.dxc[o] * p.pool_scale : 0.f;
        dcb[r][v] = ok ? p.dxc[dstride + o] * p.pool_scale : 0.f;
      }
    }
    const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
    const T* dob_b = (const T*)p.dob_in + (size_t)b * g.L * p.d_in;
    T* dxz_b = (T*)p.dxz + (size_t)b * g.L * 2 * p.d_in;
    for (int j0 = 0; j0 < g.cols; j0 += TJ) {
      float x[TJ + 6][VEC], dov[TJ + 6][VEC];
      load_x_tile<T, VEC, TJ, 3>(xz_b, g, p.d_in, i, j0, c0, act, x);
      bool tv[TJ + 6];
      int rsel[TJ + 6];
#pragma unroll
      for (int k = 0; k < TJ + 6; ++k) {
        int j = j0 - 3 + k;
        int s = i * g.cols + j;
        tv[k] = s >= 0 && s < g.L && j < g.cols + 3;
        rsel[k] = j < 0 ? 0 : (j >= g.cols ? 2 : 1);
        if (tv[k] && act) {
          VecIO<T, VEC>::load(dob_b + (size_t)tok_mem(g, s) * p.d_in + c0, dov[k]);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dov[k][v] = 0.f;
        }
      }
      // d(pre-activation): forward dir for tile indices 3..TJ+5, backward dir for 0..TJ+2
      float dpf[TJ + 6][VEC], dpb[TJ + 6][VEC];
#pragma unroll
      for (int k = 0; k < TJ + 6; ++k) {
        const bool wf_ = k >= 3, wb_ = k <= TJ + 2;
        float pf[VEC], pb[VEC];
        conv_pre<VEC, TJ>(cp, x, k, pf, pb, wf_, wb_);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float cf = rsel[k] == 0 ? dcf[0][v] : (rsel[k] == 1 ? dcf[1][v] : dcf[2][v]);
          float cb = rsel[k] == 0 ? dcb[0][v] : (rsel[k] == 1 ? dcb[1][v] : dcb[2][v]);
          dpf[k][v] = (wf_ && tv[k]) ? (Dfh[v] * dov[k][v] + cf) * fv_silu_grad(pf[v]) : 0.f;
          dpb[k][v] = (wb_ && tv[k]) ? (Dbh[v] * dov[k][v] + cb) * fv_silu_grad(pb[v]) : 0.f;
        }
      }
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) {
        const int k = jj + 3;
        if (j0 + jj < g.cols) {
          float dx[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float acc = 0.f;
#pragma unroll
            for (int kk = 0; kk < CW; ++kk) {
              acc = fmaf(cp.wf[v][kk], dpf[k + 3 - kk][v], acc);
              acc = fmaf(cp.wb[v][kk], dpb[k - 3 + kk][v], acc);
              a_wf[v][kk] = fmaf(dpf[k][v], x[k - 3 + kk][v], a_wf[v][kk]);
              a_wb[v][kk] = fmaf(dpb[k][v], x[k + 3 - kk][v], a_wb[v][kk]);
            }
            a_bf[v] += dpf[k][v];
            a_bb[v] += dpb[k][v];
            dx[v] = acc;
          }
          if (act) VecIO<T, VEC>::store(dxz_b + (size_t)tok_mem(g, i * g.cols + j0 + jj) * 2 * p.d_in + c0, dx);
        }
      }
    }
  }
  if (act) {
    float* dst = p.part + ((size_t)blockIdx.x * p.d_in + c0) * 10;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
      for (int k = 0; k < CW; ++k) {
        dst[v * 10 + k] = a_wf[v][k];
        dst[v * 10 + 4 + k] = a_wb[v][k];
      }
      dst[v * 10 + 8] = a_bf[v];
      dst[v * 10 + 9] = a_bb[v];
    }
  }
}

// ------------------------------------------------------------------ scan backward
struct ScanBwdParams {
  const void* xc;        // (2, B, Lc, d_in)
  const void* xdbl;      // (2, B*Lc, R+2N)
  const float* Wdt[2];
  const float* dtb[2];
  const float* Alog[2];
  const float* dyc;      // (B, Lc, d_in)  gradient wrt the scan output (same for both directions)
  float* dxc;            // (2, B, Lc, d_in)  gradient wrt xc through the scan (u path)
  float* dxdbl;          // (nchunks, 2, B*Lc, R+2N) reduced over the chunk's channels
  float* ckpt;           // (2, B, nseg, d_in, N) states entering each segment
  float* pA;             // (B, 2, d_in, N)   dA_log partials
  float* pW;             // (B, 2, d_in, R)   d dt_proj.weight partials
  float* pb;             // (B, 2, d_in)      d dt_proj.bias partials
  int B, Lc, d_in, R;
};

// Reduce PV per-lane values over the 64 lanes of the wave ("reduce-scatter" butterfly): on
// return lane l holds the totals of value indices [l*PV/64, (l+1)*PV/64) in v[0..PV/64).
template <int PV>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[PV], int lane) {
#pragma unroll
  for (int off = 32, h = PV / 2; off >= 1; off >>= 1, h >>= 1) {
    const bool up = lane & off;
#pragma unroll
    for (int e = 0; e < h; ++e) {
      float keep = up ? v[e + h] : v[e];
      float send = up ? v[e] : v[e + h];
      v[e] = keep + __shfl_xor(send, off, 64);
    }
  }
}

template <typename T, int N, int RMAX, int PV>
__global__ __launch_bounds__(512) void scan_cl_bwd_kernel(ScanBwdParams p) {
  constexpr int K = 4;          // steps per recompute segment
  constexpr int Q = PV / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int W = p.R + 2 * N;
  float* s_dbl = smem;                         // Lc * W
  float* s_part = smem + p.Lc * W;             // K * NW * PV
  const int dir = blockIdx.z, b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6;
  const int d = chunk * blockDim.x + tid;
  const bool act = d < p.d_in;
  const int dd = act ? d : 0;
  const T* dbl = (const T*)p.xdbl + ((size_t)dir * p.B + b) * p.Lc * W;
  for (int e = tid; e < p.Lc * W; e += blockDim.x) s_dbl[e] = io<T>::ld(dbl + e);
  __syncthreads();

  float A2[N], Araw[N], st[N], wdt[RMAX];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    Araw[n] = -__expf(p.Alog[dir][(size_t)dd * N + n]);
    A2[n] = Araw[n] * FV_LOG2E;
    st[n] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r) wdt[r] = (r < p.R) ? p.Wdt[dir][(size_t)dd * p.R + r] : 0.f;
  const float bias = p.dtb[dir][dd];
  const int nseg = (p.Lc + K - 1) / K;
  const T* u = (const T*)p.xc + ((size_t)dir * p.B + b) * p.Lc * p.d_in + dd;
  float* ck = p.ckpt + (((size_t)dir * p.B + b) * nseg * p.d_in + dd) * N;
  const size_t ck_seg = (size_t)p.d_in * N;

  auto delta_raw = [&](const float* row) {
    float dt = bias;
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < p.R) dt = fmaf(wdt[r], row[r], dt);
    return dt;
  };

  // ---- pass 1: forward sweep, checkpoint the state entering every segment but the first
  for (int step = 0; step < (nseg - 1) * K; ++step) {
    const int l = dir ? p.Lc - 1 - step : step;
    const float* row = s_dbl + l * W;
    const float dt = fv_softplus(delta_raw(row));
    const float du = dt * io<T>::ld(u + (size_t)l * p.d_in);
#pragma unroll
    for (int n = 0; n < N; ++n) st[n] = fmaf(fv_exp2(dt * A2[n]), st[n], du * row[p.R + n]);
    if ((step + 1) % K == 0 && act) {
      float* dst = ck + (size_t)((step + 1) / K) * ck_seg;
#pragma unroll
      for (int n = 0; n < N; ++n) dst[n] = st[n];
    }
  }

  // ---- pass 2: segments high-to-low
  float dxa[N], dA[N], dW[RMAX], dbias = 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) dxa[n] = dA[n] = 0.f;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) dW[r] = 0.f;

  for (int seg = nseg - 1; seg >= 0; --seg) {
    const int s0 = seg * K;
    const int ns = min(K, p.Lc - s0);
    float cur[N];
    if (seg > 0 && act) {
      const float* src = ck + (size_t)seg * ck_seg;
#pragma unroll
      for (int n = 0; n < N; ++n) cur[n] = src[n];
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n) cur[n] = 0.f;
    }
    float xs[K][N], dtr[K], dtv[K], uv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < ns) {
        const int l = dir ? p.Lc - 1 - (s0 + k) : s0 + k;
        const float* row = s_dbl + l * W;
        dtr[k] = delta_raw(row);
        dtv[k] = fv_softplus(dtr[k]);
        uv[k] = io<T>::ld(u + (size_t)l * p.d_in);
        const float du = dtv[k] * uv[k];
#pragma unroll
        for (int n = 0; n < N; ++n) {
          cur[n] = fmaf(fv_exp2(dtv[k] * A2[n]), cur[n], du * row[p.R + n]);
          xs[k][n] = cur[n];
        }
      } else {
        dtr[k] = dtv[k] = uv[k] = 0.f;
#pragma unroll
        for (int n = 0; n < N; ++n) xs[k][n] = 0.f;
      }
    }
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
      if (k < ns) {     // uniform across the block
        const int l = dir ? p.Lc - 1 - (s0 + k) : s0 + k;
        const float* row = s_dbl + l * W;
        const float gq = act ? p.dyc[((size_t)b * p.Lc + l) * p.d_in + dd] : 0.f;
        float vals[PV];
#pragma unroll
        for (int e = 0; e < PV; ++e) vals[e] = 0.f;
        float du_acc = 0.f, ddt_acc = 0.f;
        const float dtu = dtv[k] * uv[k];
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const float Bn = row[p.R + n], Cn = row[p.R + N + n];
          const float a = fv_exp2(dtv[k] * A2[n]);
          const float dx = fmaf(gq, Cn, dxa[n]);
          const float ax = xs[k][n] - dtu * Bn;          // a_t * x_{t-1}
          du_acc = fmaf(dx, Bn, du_acc);
          ddt_acc += dx * fmaf(Araw[n], ax, Bn * uv[k]);
          dA[n] = fmaf(dx * dtv[k], ax, dA[n]);
          vals[RMAX + n] = dx * dtu;                      // dB
          vals[RMAX + N + n] = gq * xs[k][n];             // dC
          dxa[n] = a * dx;
        }
        float ddraw = ddt_acc;
        if (dtr[k] <= 20.f) ddraw *= fv_sigmoid(dtr[k]);
        if (!act) ddraw = 0.f;
        dbias += ddraw;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          if (r < p.R) {
            dW[r] = fmaf(ddraw, row[r], dW[r]);
            vals[r] = ddraw * wdt[r];                     // d dt_low
          }
        }
        if (act) p.dxc[(((size_t)dir * p.B + b) * p.Lc + l) * p.d_in + d] = dtv[k] * du_acc;
        wave_reduce_scatter<PV>(vals, lane);
#pragma unroll
        for (int q = 0; q < Q; ++q) s_part[(k * NW + wv) * PV + lane * Q + q] = vals[q];
      }
    }
    __syncthreads();
    // cross-wave sum in fixed order; value slot layout: [0,RMAX) dt_low | [RMAX,+N) dB | [RMAX+N,+N) dC
    for (int e = tid; e < ns * W; e += blockDim.x) {
      const int k = e / W, c = e - k * W;
      const int slot = c < p.R ? c : (RMAX + (c - p.R));
      float t = 0.f;
      for (int w = 0; w < NW; ++w) t += s_part[(k * NW + w) * PV + slot];
      const int l = dir ? p.Lc - 1 - (s0 + k) : s0 + k;
      p.dxdbl[(((size_t)chunk * 2 + dir) * p.B + b) * p.Lc * W + (size_t)l * W + c] = t;
    }
    __syncthreads();
  }
  if (act) {
    const size_t o = ((size_t)b * 2 + dir) * p.d_in + d;
#pragma unroll
    for (int n = 0; n < N; ++n) p.pA[o * N + n] = dA[n] * Araw[n];   // A = -exp(A_log): dA_log = dA * A
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < p.R) p.pW[o * p.R + r] = dW[r];
    p.pb[o] = dbias;
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ in, float* __restrict__ out, int S, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += in[(size_t)s * n + i];
  out[i] = acc;
}

template <typename T, int VEC>
int launch_bwd_kernels(int which, const BwdParams& p, int nblocks, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 8), "mixer: d_inner %d too large for the VEC=%d row-walker", p.d_in, VEC);
  dim3 grid(nblocks), block(64 * nch);
  const bool t7 = p.geo.cols % 7 == 0;
  if (which == 0) {
    if (t7) hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 7>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((combine_bwd_kernel<T, VEC, 8>), grid, block, 0, st, p);
  } else {
    if (t7) hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 7>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_pool_bwd_kernel<T, VEC, 8>), grid, block, 0, st, p);
  }
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T>
int dispatch_bwd(int which, const BwdParams& p, int nblocks, hipStream_t st) {
  if (p.d_in % 128 == 0) return launch_bwd_kernels<T, 2>(which, p, nblocks, st);
  return launch_bwd_kernels<T, 1>(which, p, nblocks, st);
}

int check_geo_b(int B, int rows, int cols, int s_i, int s_j, int d_in, int dtype) {
  FV_CHECK(B > 0 && rows > 0 && cols > 0 && d_in > 0, "mixer: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer: dtype must be fp32 or bf16");
  FV_CHECK((s_i == cols && s_j == 1) || (s_i == 1 && s_j == rows),
           "mixer: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid", s_i, s_j, rows, cols);
  return FV_OK;
}

}  // namespace

extern "C" int fv_mixer_bwd_blocks(int batch, int rows) {
  // balanced persistent grid: every block walks the same number of rows (+-1)
  long n = (long)batch * rows;
  long per = (n + 511) / 512;
  return (int)((n + per - 1) / per);
}

extern "C" int fv_mixer_combine_bwd(const void* dg, const void* xz, const float* yc, const float* conv_w,
                                    const float* conv_b, const float* conv_w_b, const float* conv_b_b,
                                    const float* D, const float* D_b, const float* ln_w, const float* ln_b,
                                    const float* mean, const float* rstd, void* dxz, void* d_o, float* dyc,
                                    float* partials, int batch, int rows, int cols, int tok_stride_row,
                                    int tok_stride_col, int d_inner, int d_conv, int dtype, fv_stream_t stream) {
  int rc = check_geo_b(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype);
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  FV_CHECK(dg && xz && yc && conv_w && conv_w_b && D && D_b && dxz && d_o && dyc && partials,
           "mixer_combine_bwd: null pointer");
  FV_CHECK(!ln_w || (ln_b && mean && rstd), "mixer_combine_bwd: LayerNorm needs weight, bias, mean, rstd");
  BwdParams p{};
  p.dg = dg; p.xz = xz; p.yc = yc; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b;
  p.Df = D; p.Db = D_b; p.lnw = ln_w; p.lnb = ln_b; p.mean = mean; p.rstd = rstd;
  p.dxz = dxz; p.dob = d_o; p.dyc = dyc; p.part = partials;
  p.use_norm = ln_w != nullptr;
  p.geo = {rows, cols, rows * cols, tok_stride_row, tok_stride_col};
  p.B = batch; p.d_in = d_inner;
  const int nb = fv_mixer_bwd_blocks(batch, rows);
  return dtype == FV_F32 ? dispatch_bwd<float>(0, p, nb, (hipStream_t)stream)
                         : dispatch_bwd<bf16_t>(0, p, nb, (hipStream_t)stream);
}

extern "C" int fv_mixer_conv_pool_bwd(const void* xz, const void* d_o, const float* dxc, const float* conv_w,
                                      const float* conv_b, const float* conv_w_b, const float* conv_b_b,
                                      const float* D, const float* D_b, void* dxz, float* partials, int batch,
                                      int rows, int cols, int tok_stride_row, int tok_stride_col, int d_inner,
                                      int d_conv, int pool_max, float scaling_factor, int dtype,
                                      fv_stream_t stream) {
  int rc = check_geo_b(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype);
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  if (pool_max) {
    fv_set_error("mixer_conv_pool_bwd: collapse_method='max' has no backward kernel yet");
    return FV_ERR_UNSUPPORTED;
  }
  FV_CHECK(cols >= 3, "mixer_conv_pool_bwd: needs cols >= 3 (got %d)", cols);
  FV_CHECK(xz && d_o && dxc && conv_w && conv_w_b && D && D_b && dxz && partials, "mixer_conv_pool_bwd: null pointer");
  BwdParams p{};
  p.xz = xz; p.dob_in = d_o; p.dxc = dxc; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b;
  p.Df = D; p.Db = D_b; p.dxz = dxz; p.part = partials;
  p.geo = {rows, cols, rows * cols, tok_stride_row, tok_stride_col};
  p.B = batch; p.d_in = d_inner;
  p.pool_scale = scaling_factor / (float)cols;
  const int nb = fv_mixer_bwd_blocks(batch, rows);
  return dtype == FV_F32 ? dispatch_bwd<float>(1, p, nb, (hipStream_t)stream)
                         : dispatch_bwd<bf16_t>(1, p, nb, (hipStream_t)stream);
}

extern "C" int fv_mixer_scan_bwd_chunks(int d_inner) { return fv_cdiv(d_inner, 512); }

extern "C" size_t fv_mixer_scan_bwd_ckpt_floats(int batch, int Lc, int d_inner, int d_state) {
  return (size_t)2 * batch * ((Lc + 3) / 4) * d_inner * d_state;
}

extern "C" int fv_mixer_scan_bwd(const void* xc, const void* x_dbl, const float* dt_w, const float* dt_bias,
                                 const float* A_log, const float* dt_w_b, const float* dt_bias_b,
                                 const float* A_log_b, const float* dyc, float* dxc, float* dx_dbl, float* ckpt,
                                 float* pA, float* pW, float* pb, int batch, int Lc, int d_inner, int dt_rank,
                                 int d_state, int dtype, fv_stream_t stream) {
  FV_CHECK(batch > 0 && Lc > 0 && d_inner > 0 && dt_rank > 0, "mixer_scan_bwd: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer_scan_bwd: dtype must be fp32 or bf16");
  FV_CHECK(d_state == 16, "mixer_scan_bwd: only d_state == 16 is built (got %d)", d_state);
  FV_CHECK(dt_rank <= 96, "mixer_scan_bwd: dt_rank %d > 96", dt_rank);
  FV_CHECK(xc && x_dbl && dt_w && dt_bias && A_log && dt_w_b && dt_bias_b && A_log_b && dyc && dxc && dx_dbl &&
               ckpt && pA && pW && pb, "mixer_scan_bwd: null pointer");
  ScanBwdParams p{};
  p.xc = xc; p.xdbl = x_dbl; p.dyc = dyc; p.dxc = dxc; p.dxdbl = dx_dbl; p.ckpt = ckpt;
  p.pA = pA; p.pW = pW; p.pb = pb;
  p.Wdt[0] = dt_w; p.Wdt[1] = dt_w_b; p.dtb[0] = dt_bias; p.dtb[1] = dt_bias_b;
  p.Alog[0] = A_log; p.Alog[1] = A_log_b;
  p.B = batch; p.Lc = Lc; p.d_in = d_inner; p.R = dt_rank;
  const int nchunks = fv_mixer_scan_bwd_chunks(d_inner);
  const int per = fv_cdiv(d_inner, nchunks);
  const int bs = fv_cdiv(per, 64) * 64;
  const int NW = bs / 64;
  const int W = dt_rank + 2 * d_state;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nchunks, batch, 2), block(bs);
#define FV_LAUNCH_SB(TT, RM, PVV)                                                           \
  do {                                                                                      \
    size_t smem = ((size_t)Lc * W + (size_t)4 * NW * PVV) * 4;                              \
    FV_CHECK(smem <= 64 * 1024, "mixer_scan_bwd: pooled length %d too long for the LDS stage", Lc); \
    hipLaunchKernelGGL((scan_cl_bwd_kernel<TT, 16, RM, PVV>), grid, block, smem, st, p);    \
  } while (0)
  if (dtype == FV_F32) {
    if (dt_rank <= 12) FV_LAUNCH_SB(float, 12, 64); else if (dt_rank <= 24) FV_LAUNCH_SB(float, 24, 64);
    else if (dt_rank <= 48) FV_LAUNCH_SB(float, 48, 128); else FV_LAUNCH_SB(float, 96, 128);
  } else {
    if (dt_rank <= 12) FV_LAUNCH_SB(bf16_t, 12, 64); else if (dt_rank <= 24) FV_LAUNCH_SB(bf16_t, 24, 64);
    else if (dt_rank <= 48) FV_LAUNCH_SB(bf16_t, 48, 128); else FV_LAUNCH_SB(bf16_t, 96, 128);
  }
#undef FV_LAUNCH_SB
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_reduce_partials(const float* partials, float* out, int n_partials, size_t n,
                                  fv_stream_t stream) {
  FV_CHECK(partials && out && n_partials > 0, "reduce_partials: bad arguments");
  if (n == 0) return FV_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(fv_cdiv((long)n, 256)), dim3(256), 0, (hipStream_t)stream,
                     partials, out, n_partials, n);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
