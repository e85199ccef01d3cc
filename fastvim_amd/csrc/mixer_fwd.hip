// hipcc-flags: -fgpu-flush-denormals-to-zero
// Forward kernels of the fused FastVim mixer "middle" (everything between the in_proj GEMM
// output xz and the out_proj GEMM input), channel-last.  Replaces, for one bidirectional
// block, the ~35 launches of mamba_simple_faster.py:270-444:
//
//   fv_mixer_conv_pool_fwd : both causal (forward dir) and anti-causal (backward dir)
//                            depthwise conv + SiLU, the mean/max pooling over `cols`
//                            (mamba_simple_faster.py:272-305) and the D-weighted skip term
//                            skip = D*conv_f + D_b*conv_b of (:356-358, :412-416) -- one read of x.
//   (fv_mixer_scan_fwd, the dt_proj + scan over the pooled rows, lives in scan_cl.hip)
//   fv_mixer_combine_fwd   : expand the scan output over `cols`, + skip, average the two directions,
//                            LayerNorm over d_in, * SiLU(z)  (:434-441) -- reads skip, z once, writes once.
//
// The flip()s of the reference never happen: the backward direction is the anti-causal
// conv on the original order and a scan over pooled rows in descending order.
//
// These launches are short (tens of microseconds) and instruction-issue bound, not HBM bound: a SiLU
// costs two quarter-rate transcendentals, so every conv+SiLU is evaluated once in the forward pass
// (conv_pool) and its D-weighted sum handed to combine as `skip` instead of being recomputed there.
#include <stdlib.h>

#include "mixer_common.h"

namespace {

using fvi::FwdParams;

template <int VEC>
__device__ __forceinline__ void conv_silu_both(const ChanParams<VEC>& cp, const float (&xw)[7][VEC],
                                               float (&xf)[VEC], float (&xb)[VEC]) {
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float pf = cp.bf[v], pb = cp.bb[v];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
      pf = fmaf(cp.wf[v][k], xw[k][v], pf);          // x[s-3+k]
      pb = fmaf(cp.wb[v][k], xw[6 - k][v], pb);      // x[s+3-k]
    }
    xf[v] = fv_silu(pf);
    xb[v] = fv_silu(pb);
  }
}

// ------------------------------------------------------------------ conv + pool (+ skip), generic tiles
template <typename T, int VEC, int TJ, bool TP, bool PMAX>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void conv_pool_fwd_kernel(FwdParams p) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x, b = blockIdx.y;
  const int c0 = (wv * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  float Df[VEC], Db[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    Df[v] = act && p.skip ? p.Df[c0 + v] : 0.f;
    Db[v] = act && p.skip ? p.Db[c0 + v] : 0.f;
  }
  const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
  T* sk_b = (T*)p.skip + (size_t)b * g.L * p.d_in;
  // pooled accumulators: registers for tpp == 1; thread-private LDS slots [dir][slot][thread][VEC] otherwise
  extern __shared__ __attribute__((aligned(16))) float s_pool[];
  const int tpp = TP ? g.tpp : 1, nthr = blockDim.x;
  float accf[VEC], accb[VEC];
  float argf[VEC], argb[VEC];            // PMAX: column of the (first) maximum, for the backward pass
  const float init = PMAX ? -INFINITY : 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { accf[v] = accb[v] = init; argf[v] = argb[v] = 0.f; }
  // tpp > 1: slots [0, 2 tpp) hold the pooled values, PMAX adds [2 tpp, 4 tpp) for the argmax columns
  // pcols == 1: nothing is pooled (every token is its own group -- the un-pooled Vim mixer runs as rows x 1 x tpp):
  // the conv output goes straight to xc, no accumulators
  const bool nopool = TP && g.pcols == 1;
  if constexpr (TP)
    if (!nopool)
      for (int c = 0; c < (PMAX ? 4 : 2) * tpp; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) s_pool[(c * nthr + threadIdx.x) * VEC + v] = c < 2 * tpp ? init : 0.f;
  for (int j0 = 0; j0 < g.cols; j0 += TJ) {
    float x[TJ + 6][VEC];
    load_x_tile<T, VEC, TJ, 3, TP>(xz_b, g, p.d_in, i, j0, c0, act, x);
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      if (j0 + jj < g.cols) {
        float xw[7][VEC], xf[VEC], xb[VEC];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
          for (int v = 0; v < VEC; ++v) xw[k][v] = x[jj + k][v];
        conv_silu_both<VEC>(cp, xw, xf, xb);
        if (p.skip && act) {
          float sk[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) sk[v] = fmaf(Df[v], xf[v], Db[v] * xb[v]);
          VecIO<T, VEC>::store(sk_b + (size_t)tok_mem<TP>(g, i * g.cols + j0 + jj) * p.d_in + c0, sk);
        }
        if constexpr (!TP) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if constexpr (PMAX) {
              const float col = (float)(j0 + jj);
              if (xf[v] > accf[v]) { accf[v] = xf[v]; argf[v] = col; }
              if (xb[v] > accb[v]) { accb[v] = xb[v]; argb[v] = col; }
            } else {
              accf[v] += xf[v];
              accb[v] += xb[v];
            }
          }
        } else if (nopool) {
          if (act) {
            const size_t o = (((size_t)b * g.rows + i) * tpp + (j0 + jj)) * p.d_in + c0;
            const size_t dstr = (size_t)p.B * g.rows * tpp * p.d_in;
            float of[VEC], ob[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { of[v] = xf[v] * p.pool_scale; ob[v] = xb[v] * p.pool_scale; }
            VecIO<T, VEC>::store((T*)p.xc + o, of);
            VecIO<T, VEC>::store((T*)p.xc + dstr + o, ob);
            if constexpr (PMAX) {
              if (p.amax) {
                float zero[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) zero[v] = 0.f;
                VecIO<T, VEC>::store((T*)p.amax + o, zero);
                VecIO<T, VEC>::store((T*)p.amax + dstr + o, zero);
              }
            }
          }
        } else {
          const int slot = (j0 + jj) % tpp;
          float* af = s_pool + (slot * nthr + threadIdx.x) * VEC;
          float* ab = s_pool + ((tpp + slot) * nthr + threadIdx.x) * VEC;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if constexpr (PMAX) {
              const float col = (float)((j0 + jj) / tpp);
              float* gf = af + (size_t)2 * tpp * nthr * VEC;
              float* gb = ab + (size_t)2 * tpp * nthr * VEC;
              if (xf[v] > af[v]) { af[v] = xf[v]; gf[v] = col; }
              if (xb[v] > ab[v]) { ab[v] = xb[v]; gb[v] = col; }
            } else {
              af[v] += xf[v];
              ab[v] += xb[v];
            }
          }
        }
      }
    }
  }
  if (act && !nopool) {
    T* xc = (T*)p.xc;
    const size_t dstride = (size_t)p.B * g.rows * tpp * p.d_in;
    for (int c = 0; c < tpp; ++c) {
      if constexpr (TP) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          accf[v] = s_pool[(c * nthr + threadIdx.x) * VEC + v];
          accb[v] = s_pool[((tpp + c) * nthr + threadIdx.x) * VEC + v];
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        accf[v] *= p.pool_scale;
        accb[v] *= p.pool_scale;
      }
      const size_t o = (((size_t)b * g.rows + i) * tpp + c) * p.d_in + c0;
      VecIO<T, VEC>::store(xc + o, accf);
      VecIO<T, VEC>::store(xc + dstride + o, accb);
      if constexpr (PMAX) {
        if (p.amax) {
          if constexpr (TP) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              argf[v] = s_pool[((2 * tpp + c) * nthr + threadIdx.x) * VEC + v];
              argb[v] = s_pool[((3 * tpp + c) * nthr + threadIdx.x) * VEC + v];
            }
          }
          VecIO<T, VEC>::store((T*)p.amax + o, argf);
          VecIO<T, VEC>::store((T*)p.amax + dstride + o, argb);
        }
      }
    }
  }
}

// ------------------------------------------------------------------ combine: expand + skip + LayerNorm + gate
// Purely per-token once `skip` exists.  Block = RG row groups x NCH channel-chunk waves, persistent over
// the pooling rows; a row group walks one row TT tokens at a time with the next TT tokens' loads in flight.
constexpr int RGMAXF = 4;

template <typename T, int VEC, int TT, bool TP>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void combine_fwd_kernel(FwdParams p, int nch, int RG) {
  __shared__ float s_red[RGMAXF * TT * 16];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wv / nch, cw = wv - rg * nch;
  const int c0 = (cw * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  const int tpp = TP ? g.tpp : 1;
  float lw[VEC], lb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    lw[v] = act && p.use_norm ? p.lnw[c0 + v] : 1.f;
    lb[v] = act && p.use_norm ? p.lnb[c0 + v] : 0.f;
  }
  const float inv_d = 1.f / (float)p.d_in;
  const size_t ydir = (size_t)p.B * g.rows * tpp * p.d_in;
  const int nrows = p.B * g.rows;
  const int nit = (nrows + gridDim.x * RG - 1) / (gridDim.x * RG);
  for (int it = 0; it < nit; ++it) {
    const int row = (it * gridDim.x + blockIdx.x) * RG + rg;
    const bool rv = row < nrows;                         // uniform per row group
    const int b = rv ? row / g.rows : 0, i = rv ? row - b * g.rows : 0;
    const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in + p.d_in + c0;   // z half
    const T* sk_b = (const T*)p.skip + (size_t)b * g.L * p.d_in + c0;
    T* g_b = (T*)p.g + (size_t)b * g.L * p.d_in + c0;
    const float* yc_r = p.yc + (size_t)(rv ? row : 0) * tpp * p.d_in + (act ? c0 : 0);
    float ysum[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) ysum[v] = (!TP && rv && act) ? yc_r[v] + yc_r[ydir + v] : 0.f;
    RawVec<T, VEC> n_sk[TT], n_z[TT];
    auto fetch = [&](int j0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (rv && act && j0 + t < g.cols) {
          const int m = tok_mem<TP>(g, i * g.cols + j0 + t);
          n_sk[t].load(sk_b + (size_t)m * p.d_in);
          n_z[t].load(xz_b + (size_t)m * 2 * p.d_in);
        } else {
          n_sk[t].zero(); n_z[t].zero();
        }
      }
    };
    fetch(0);
    for (int j0 = 0; j0 < g.cols; j0 += TT) {
      float o[TT][VEC], zq[TT][VEC], s1[TT], mean[TT], rstd[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        n_sk[t].get(o[t]);
        n_z[t].get(zq[t]);
      }
      if (j0 + TT < g.cols) fetch(j0 + TT);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if constexpr (TP) {
          if (rv && act && j0 + t < g.cols) {
            const float* y = yc_r + (size_t)((j0 + t) % tpp) * p.d_in;
#pragma unroll
            for (int v = 0; v < VEC; ++v) ysum[v] = y[v] + y[ydir + v];
          }
        }
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          o[t][v] = act ? 0.5f * (ysum[v] + o[t][v]) : 0.f;
          acc += o[t][v];
        }
        s1[t] = acc;
      }
      if (p.use_norm) {
        // mean over d_in, then the centred second moment (two exact passes over registers)
#pragma unroll
        for (int t = 0; t < TT; ++t) s1[t] = wave_sum_uniform(s1[t]);
        if (nch > 1) {
          __syncthreads();
          if (lane == 0)
#pragma unroll
            for (int t = 0; t < TT; ++t) s_red[(rg * TT + t) * 16 + cw] = s1[t];
          __syncthreads();
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            float a = 0.f;
            for (int w = 0; w < nch; ++w) a += s_red[(rg * TT + t) * 16 + w];
            s1[t] = a;
          }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          mean[t] = s1[t] * inv_d;
          float acc = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float c = act ? o[t][v] - mean[t] : 0.f;
            acc += c * c;
          }
          s1[t] = wave_sum_uniform(acc);
        }
        if (nch > 1) {
          __syncthreads();
          if (lane == 0)
#pragma unroll
            for (int t = 0; t < TT; ++t) s_red[(rg * TT + t) * 16 + cw] = s1[t];
          __syncthreads();
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            float a = 0.f;
            for (int w = 0; w < nch; ++w) a += s_red[(rg * TT + t) * 16 + w];
            s1[t] = a;
          }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) rstd[t] = rsqrtf(s1[t] * inv_d + p.eps);
      } else {
#pragma unroll
        for (int t = 0; t < TT; ++t) { mean[t] = 0.f; rstd[t] = 1.f; }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (rv && j0 + t < g.cols) {
          const int m = tok_mem<TP>(g, i * g.cols + j0 + t);
          if (act) {
            float out[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v)
              out[v] = ((o[t][v] - mean[t]) * rstd[t] * lw[v] + lb[v]) * fv_silu(zq[t][v]);
            VecIO<T, VEC>::store(g_b + (size_t)m * p.d_in, out);
          }
          if (p.use_norm && cw == 0 && lane == 0) {
            p.mean[(size_t)b * g.L + m] = mean[t];
            p.rstd[(size_t)b * g.L + m] = rstd[t];
          }
        }
      }
    }
  }
}

int persistent_blocks_f(long nrows, int rg) {
  long groups = (nrows + rg - 1) / rg;
  long per = (groups + 511) / 512;
  return (int)((groups + per - 1) / per);
}

template <typename T, int VEC>
int launch_conv_pool(const FwdParams& p, int pool_max, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 8), "mixer: d_inner %d too large for the VEC=%d row-walker", p.d_in, VEC);
  dim3 grid(p.geo.rows, p.B), block(64 * nch);
  const bool tp = p.geo.tpp > 1;
  const size_t smem = (tp && p.geo.pcols > 1) ? (size_t)(pool_max ? 4 : 2) * p.geo.tpp * 64 * nch * VEC * 4 : 0;
  FV_CHECK(smem <= 160 * 1024, "mixer_conv_pool_fwd: tokens_per_patch %d too large", p.geo.tpp);
  if (smem > 64 * 1024) {     // opt in to > 64 KiB of dynamic LDS (once per instantiation; not a stream operation)
    static FvOncePerDevice done;   
    if (done.first()) {
      (void)hipFuncSetAttribute((const void*)conv_pool_fwd_kernel<T, VEC, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_pool_fwd_kernel<T, VEC, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)0;     
    }
  }
#define FV_CP(K, ...)                                                                        \
  do {                                                                                       \
    if (pool_max) hipLaunchKernelGGL((K<__VA_ARGS__, true>), grid, block, smem, st, p);      \
    else hipLaunchKernelGGL((K<__VA_ARGS__, false>), grid, block, smem, st, p);              \
  } while (0)
  static const bool rowk = (fv_tune("FASTVIM_FWD_ROWK", 1) != 0);   // tuning hook
  if (rowk && !pool_max) {     // short rows / 8-token cells, mean pooling: the packed-math kernels (convpool_fwd_row.hip): the whole-row packed-math kernel (convpool_fwd_row.hip)
    int rc = fvi::conv_pool_fwd_row(p, pool_max, sizeof(T) == 4 ? FV_F32 : FV_BF16, st);
    if (rc != FV_ERR_UNSUPPORTED) return rc;
  }
  if (tp) FV_CP(conv_pool_fwd_kernel, T, VEC, 8, true);
  else if (p.geo.cols % 7 == 0) FV_CP(conv_pool_fwd_kernel, T, VEC, 7, false);
  else FV_CP(conv_pool_fwd_kernel, T, VEC, 8, false);
#undef FV_CP
  FV_LAUNCH_CHECK();
  return FV_OK;
}

int rg_combine_f(int d_in, int VEC) { int nch = fv_cdiv(d_in, 64 * VEC); int r = 8 / nch; return r < 1 ? 1 : (r > RGMAXF ? RGMAXF : r); }
int vec_combine_f(int d_in) { return (d_in % 384 == 0 && d_in <= 8 * 384) ? 6 : (d_in % 256 == 0 && d_in <= 8 * 256) ? 4 : (d_in % 128 == 0 && d_in <= 8 * 128) ? 2 : 1; }

template <typename T, int VEC>
int launch_combine(const FwdParams& p, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 8), "mixer: d_inner %d too large for the VEC=%d row-walker", p.d_in, VEC);
  const int rg = rg_combine_f(p.d_in, VEC);
  dim3 grid(persistent_blocks_f((long)p.B * p.geo.rows, rg)), block(64 * nch * rg);
  const bool tp = p.geo.tpp > 1;
  if (p.geo.cols % 2 == 0) {
    if (tp) hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 2, true>), grid, block, 0, st, p, nch, rg);
    else hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 2, false>), grid, block, 0, st, p, nch, rg);
  } else {
    if (tp) hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 1, true>), grid, block, 0, st, p, nch, rg);
    else hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 1, false>), grid, block, 0, st, p, nch, rg);
  }
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T>
int dispatch_fwd(int which, const FwdParams& p, int pool_max, hipStream_t st) {
  if (which == 1) {
    {
      int rc = fvi::combine_fwd_wave(p, sizeof(T) == 4 ? FV_F32 : FV_BF16, st);
      if (rc != FV_ERR_UNSUPPORTED) return rc;
    }
    const int v = vec_combine_f(p.d_in);
    if (v == 6) return launch_combine<T, 6>(p, st);
    if (v == 4) return launch_combine<T, 4>(p, st);
    if (v == 2) return launch_combine<T, 2>(p, st);
    return launch_combine<T, 1>(p, st);
  }
  static const int force = fv_tune("FASTVIM_FWD_VEC", 0);   // tuning hook
  if ((force == 2 || p.geo.tpp > 1) && p.d_in % 128 == 0 && p.d_in <= 8 * 128) return launch_conv_pool<T, 2>(p, pool_max, st);
  if (p.geo.tpp > 1 && p.d_in % 256 == 0 && p.d_in <= 8 * 256) return launch_conv_pool<T, 4>(p, pool_max, st);
  if (p.geo.tpp > 1) return launch_conv_pool<T, 1>(p, pool_max, st);
  if (p.d_in % 384 == 0) return launch_conv_pool<T, 6>(p, pool_max, st);
  if (p.d_in % 256 == 0) return launch_conv_pool<T, 4>(p, pool_max, st);
  return launch_conv_pool<T, 1>(p, pool_max, st);
}

int check_geo(int B, int rows, int cols, int s_i, int s_j, int d_in, int dtype, int tpp) {
  FV_CHECK(B > 0 && rows > 0 && cols > 0 && d_in > 0 && tpp > 0, "mixer: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer: dtype must be fp32 or bf16");
  FV_CHECK((s_i == cols && s_j == 1) || (s_i == 1 && s_j == rows),
           "mixer: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid", s_i, s_j, rows, cols);
  return FV_OK;
}

}  // namespace

extern "C" int fv_mixer_conv_pool_fwd(const void* xz, const float* conv_w, const float* conv_b,
                                      const float* conv_w_b, const float* conv_b_b, const float* D,
                                      const float* D_b, void* xc, void* skip, void* amax, int batch, int rows, int cols,
                                      int tok_stride_row, int tok_stride_col, int tokens_per_patch, int d_inner,
                                      int d_conv, int pool_max, float scaling_factor, int dtype,
                                      fv_stream_t stream) {
  int rc = check_geo(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype, tokens_per_patch);
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  FV_CHECK(xz && conv_w && conv_w_b && xc, "mixer_conv_pool_fwd: null pointer");
  FV_CHECK(!skip || (D && D_b), "mixer_conv_pool_fwd: skip output needs D and D_b");
  FwdParams p{};
  p.xz = xz; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b; p.xc = xc;
  p.Df = D; p.Db = D_b; p.skip = skip; p.amax = pool_max ? amax : nullptr;
  FV_CHECK(!(pool_max && amax) || cols <= 256, "mixer_conv_pool_fwd: argmax columns are stored in the activation dtype (cols <= 256)");
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner;
  p.pool_scale = pool_max ? 1.f : scaling_factor / (float)cols;
  return dtype == FV_F32 ? dispatch_fwd<float>(0, p, pool_max, (hipStream_t)stream)
                         : dispatch_fwd<bf16_t>(0, p, pool_max, (hipStream_t)stream);
}

extern "C" int fv_mixer_combine_fwd(const void* xz, const void* skip, const float* yc, const float* ln_w,
                                    const float* ln_b, float ln_eps, void* g, float* mean, float* rstd,
                                    int batch, int rows, int cols, int tok_stride_row, int tok_stride_col,
                                    int tokens_per_patch, int d_inner, int dtype, fv_stream_t stream) {
  int rc = check_geo(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype, tokens_per_patch);
  if (rc) return rc;
  FV_CHECK(xz && skip && yc && g, "mixer_combine_fwd: null pointer");
  FV_CHECK(!ln_w || (ln_b && mean && rstd), "mixer_combine_fwd: LayerNorm needs weight, bias, mean, rstd");
  FwdParams p{};
  p.xz = xz; p.skip = const_cast<void*>(skip); p.yc = yc;
  p.lnw = ln_w; p.lnb = ln_b; p.eps = ln_eps; p.g = g; p.mean = mean; p.rstd = rstd;
  p.use_norm = ln_w != nullptr;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner;
  return dtype == FV_F32 ? dispatch_fwd<float>(1, p, 0, (hipStream_t)stream)
                         : dispatch_fwd<bf16_t>(1, p, 0, (hipStream_t)stream);
}
