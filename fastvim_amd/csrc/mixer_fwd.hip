// Forward kernels of the fused FastVim mixer "middle" (everything between the in_proj GEMM
// output xz and the out_proj GEMM input), channel-last.  Replaces, for one bidirectional
// block, the ~35 launches of mamba_simple_faster.py:270-444:
//
//   fv_mixer_conv_pool_fwd : both causal (forward dir) and anti-causal (backward dir)
//                            depthwise conv + SiLU and the mean/max pooling over `cols`
//                            (mamba_simple_faster.py:272-305) -- one read of x.
//   (fv_mixer_scan_fwd, the dt_proj + scan over the pooled rows, lives in scan_cl.hip)
//   fv_mixer_combine_fwd   : recompute conv, expand the scan output over `cols`, + D*x skip,
//                            average the two directions, LayerNorm over d_in, * SiLU(z)
//                            (:356-358, :412-416, :434-441) -- reads x,z once, writes once.
//
// The flip()s of the reference never happen: the backward direction is the anti-causal
// conv on the original order and a scan over pooled rows in descending order.
#include <stdlib.h>

#include "mixer_common.h"

namespace {

struct FwdParams {
  const void* xz;                       // (B, L, 2*d_in)
  const float *wf, *bf, *wb, *bb;       // conv1d / conv1d_b: (d_in, CW), (d_in)
  void* xc;                             // (2, B, rows, d_in) pooled conv output [dir 0 = fwd]
  const float* yc;                      // (2, B, rows, d_in) scan output
  const float *Df, *Db, *lnw, *lnb;     // (d_in)
  void* g;                              // (B, L, d_in) gated LayerNorm output
  void* xhat;                           // (B, L, d_in) normalised pre-gate value, saved for backward
  float *mean, *rstd;                   // (B*L) LayerNorm statistics (saved for backward)
  Geo geo;
  int B, d_in;
  int pool_max;
  float pool_scale;                     // scaling_factor / cols (mean) or 1 (max)
  float eps;
  int use_norm;
};

// conv pre-activations of token jj of the tile (x index jj+3 is the token itself)
template <int VEC, int TJ>
__device__ __forceinline__ void conv_both(const ChanParams<VEC>& cp, const float (&x)[TJ + 6][VEC], int jj,
                                          float (&xf)[VEC], float (&xb)[VEC]) {
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    float pf = cp.bf[v], pb = cp.bb[v];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
      pf = fmaf(cp.wf[v][k], x[jj + k][v], pf);             // x[s-3+k]
      pb = fmaf(cp.wb[v][k], x[jj + 2 * (CW - 1) - k][v], pb);  // x[s+3-k]
    }
    xf[v] = fv_silu(pf);
    xb[v] = fv_silu(pb);
  }
}

// ------------------------------------------------------------------ conv + pool
template <typename T, int VEC, int TJ, bool TP>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void conv_pool_fwd_kernel(FwdParams p) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x, b = blockIdx.y;
  const int c0 = (wv * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
  // pooled accumulators: registers for tpp == 1; thread-private LDS slots [dir][slot][thread][VEC] otherwise
  extern __shared__ __attribute__((aligned(16))) float s_pool[];
  const int tpp = TP ? g.tpp : 1, nthr = blockDim.x;
  float accf[VEC], accb[VEC];
  const float init = p.pool_max ? -INFINITY : 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) accf[v] = accb[v] = init;
  if constexpr (TP)
    for (int c = 0; c < 2 * tpp; ++c)
#pragma unroll
      for (int v = 0; v < VEC; ++v) s_pool[(c * nthr + threadIdx.x) * VEC + v] = init;
  for (int j0 = 0; j0 < g.cols; j0 += TJ) {
    float x[TJ + 6][VEC];
    load_x_tile<T, VEC, TJ, 3, TP>(xz_b, g, p.d_in, i, j0, c0, act, x);
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      if (j0 + jj < g.cols) {
        float xf[VEC], xb[VEC];
        conv_both<VEC, TJ>(cp, x, jj, xf, xb);
        if constexpr (!TP) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            accf[v] = p.pool_max ? fmaxf(accf[v], xf[v]) : accf[v] + xf[v];
            accb[v] = p.pool_max ? fmaxf(accb[v], xb[v]) : accb[v] + xb[v];
          }
        } else {
          const int slot = (j0 + jj) % tpp;
          float* af = s_pool + (slot * nthr + threadIdx.x) * VEC;
          float* ab = s_pool + ((tpp + slot) * nthr + threadIdx.x) * VEC;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            af[v] = p.pool_max ? fmaxf(af[v], xf[v]) : af[v] + xf[v];
            ab[v] = p.pool_max ? fmaxf(ab[v], xb[v]) : ab[v] + xb[v];
          }
        }
      }
    }
  }
  if (act) {
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
    }
  }
}

// ------------------------------------------------------------------ combine
template <typename T, int VEC, int TJ, bool TP>
__global__ __launch_bounds__(VEC == 1 ? 1024 : 512) void combine_fwd_kernel(FwdParams p) {
  __shared__ float s_red[16 * TJ];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int i = blockIdx.x, b = blockIdx.y;
  const int c0 = (wv * 64 + lane) * VEC;
  const bool act = c0 < p.d_in;
  const Geo g = p.geo;
  ChanParams<VEC> cp;
  cp.load(p.wf, p.bf, p.wb, p.bb, c0, act);
  float Df[VEC], Db[VEC], lw[VEC], lb[VEC], ysum[VEC];
  const int tpp = TP ? g.tpp : 1;
  const size_t yrow = ((size_t)b * g.rows + i) * tpp;              // first pooled index of this row
  const size_t ydir = (size_t)p.B * g.rows * tpp * p.d_in;
  {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      Df[v] = act ? p.Df[c0 + v] : 0.f;
      Db[v] = act ? p.Db[c0 + v] : 0.f;
      lw[v] = act && p.use_norm ? p.lnw[c0 + v] : 1.f;
      lb[v] = act && p.use_norm ? p.lnb[c0 + v] : 0.f;
      ysum[v] = act ? p.yc[yrow * p.d_in + c0 + v] + p.yc[ydir + yrow * p.d_in + c0 + v] : 0.f;   // scan outputs, both dirs
    }
  }
  const T* xz_b = (const T*)p.xz + (size_t)b * g.L * 2 * p.d_in;
  T* g_b = (T*)p.g + (size_t)b * g.L * p.d_in;
  const float inv_d = 1.f / (float)p.d_in;
  for (int j0 = 0; j0 < g.cols; j0 += TJ) {
    float x[TJ + 6][VEC];
    load_x_tile<T, VEC, TJ, 3, TP>(xz_b, g, p.d_in, i, j0, c0, act, x);
    RawVec<T, VEC> zr[TJ];           // gate inputs fetched with the tile, not at their use
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      if (act && j0 + jj < g.cols)
        zr[jj].load(xz_b + (size_t)tok_mem<TP>(g, i * g.cols + j0 + jj) * 2 * p.d_in + p.d_in + c0);
      else
        zr[jj].zero();
    }
    float o[TJ][VEC], s1[TJ];
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      float xf[VEC], xb[VEC];
      conv_both<VEC, TJ>(cp, x, jj, xf, xb);
      float acc = 0.f;
      if (TP && act) {       // channel-wise tokenization: the scan output of this token's channel slot
        const size_t yo = (yrow + (j0 + jj) % tpp) * p.d_in + c0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ysum[v] = p.yc[yo + v] + p.yc[ydir + yo + v];
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        o[jj][v] = act ? 0.5f * (ysum[v] + Df[v] * xf[v] + Db[v] * xb[v]) : 0.f;
        acc += o[jj][v];
      }
      s1[jj] = acc;
    }
    float mean[TJ], rstd[TJ];
    if (p.use_norm) {
      // mean over d_in, then centred second moment (two exact passes over registers)
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) s1[jj] = wave_sum_uniform(s1[jj]);
      if (nw > 1) {
        __syncthreads();
        if (lane == 0)
#pragma unroll
          for (int jj = 0; jj < TJ; ++jj) s_red[wv * TJ + jj] = s1[jj];
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
          float t = 0.f;
          for (int w = 0; w < nw; ++w) t += s_red[w * TJ + jj];
          s1[jj] = t;
        }
      }
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) {
        mean[jj] = s1[jj] * inv_d;
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float c = act ? o[jj][v] - mean[jj] : 0.f;
          acc += c * c;
        }
        s1[jj] = wave_sum_uniform(acc);
      }
      if (nw > 1) {
        __syncthreads();
        if (lane == 0)
#pragma unroll
          for (int jj = 0; jj < TJ; ++jj) s_red[wv * TJ + jj] = s1[jj];
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
          float t = 0.f;
          for (int w = 0; w < nw; ++w) t += s_red[w * TJ + jj];
          s1[jj] = t;
        }
      }
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) rstd[jj] = rsqrtf(s1[jj] * inv_d + p.eps);
    } else {
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) { mean[jj] = 0.f; rstd[jj] = 1.f; }
    }
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      if (j0 + jj < g.cols) {
        int m = tok_mem<TP>(g, i * g.cols + j0 + jj);
        if (act) {
          float z[VEC], out[VEC], xh[VEC];
          zr[jj].get(z);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            xh[v] = (o[jj][v] - mean[jj]) * rstd[jj];
            out[v] = (xh[v] * lw[v] + lb[v]) * fv_silu(z[v]);
          }
          VecIO<T, VEC>::store(g_b + (size_t)m * p.d_in + c0, out);
          if (p.xhat) VecIO<T, VEC>::store((T*)p.xhat + ((size_t)b * g.L + m) * p.d_in + c0, xh);
        }
        if (p.use_norm && threadIdx.x == 0) {
          p.mean[(size_t)b * g.L + m] = mean[jj];
          p.rstd[(size_t)b * g.L + m] = rstd[jj];
        }
      }
    }
  }
}

template <typename T, int VEC>
int launch_fwd_kernels(int which, const FwdParams& p, hipStream_t st) {
  const int nch = fv_cdiv(p.d_in, 64 * VEC);
  FV_CHECK(nch <= (VEC == 1 ? 16 : 8), "mixer: d_inner %d too large for the VEC=%d row-walker", p.d_in, VEC);
  dim3 grid(p.geo.rows, p.B), block(64 * nch);
  const bool t14 = p.geo.cols % 7 == 0 && p.geo.tpp == 1;
  if (which == 0) {
    const size_t smem = p.geo.tpp > 1 ? (size_t)2 * p.geo.tpp * 64 * nch * VEC * 4 : 0;
    FV_CHECK(smem <= 160 * 1024, "mixer_conv_pool_fwd: tokens_per_patch %d too large", p.geo.tpp);
    if (smem > 64 * 1024) {     // opt in to > 64 KiB of dynamic LDS (once per instantiation; not a stream operation)
      static bool done = false;
      if (!done) {
        (void)hipFuncSetAttribute((const void*)conv_pool_fwd_kernel<T, VEC, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
      }
    }
    if (p.geo.tpp > 1) hipLaunchKernelGGL((conv_pool_fwd_kernel<T, VEC, 8, true>), grid, block, smem, st, p);
    else if (t14) hipLaunchKernelGGL((conv_pool_fwd_kernel<T, VEC, 7, false>), grid, block, smem, st, p);
    else hipLaunchKernelGGL((conv_pool_fwd_kernel<T, VEC, 8, false>), grid, block, smem, st, p);
  } else {
    if (p.geo.tpp > 1) hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 8, true>), grid, block, 0, st, p);
    else if (t14) hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 7, false>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((combine_fwd_kernel<T, VEC, 8, false>), grid, block, 0, st, p);
  }
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <typename T>
int dispatch_vec(int which, const FwdParams& p, hipStream_t st) {
  static const int force = getenv("FASTVIM_FWD_VEC") ? atoi(getenv("FASTVIM_FWD_VEC")) : 0;   // tuning hook
  if ((force == 2 || p.geo.tpp > 1) && p.d_in % 128 == 0 && p.d_in <= 8 * 128) return launch_fwd_kernels<T, 2>(which, p, st);
  if (p.geo.tpp > 1 && p.d_in % 256 == 0 && p.d_in <= 8 * 256) return launch_fwd_kernels<T, 4>(which, p, st);
  if (p.geo.tpp > 1) return launch_fwd_kernels<T, 1>(which, p, st);
  if (p.d_in % 384 == 0) return launch_fwd_kernels<T, 6>(which, p, st);
  if (p.d_in % 256 == 0) return launch_fwd_kernels<T, 4>(which, p, st);
  return launch_fwd_kernels<T, 1>(which, p, st);
}

int check_geo(int B, int rows, int cols, int s_i, int s_j, int d_in, int dtype, int tpp = 1) {
  FV_CHECK(B > 0 && rows > 0 && cols > 0 && d_in > 0 && tpp > 0, "mixer: empty dimension");
  FV_CHECK(dtype == FV_F32 || dtype == FV_BF16, "mixer: dtype must be fp32 or bf16");
  FV_CHECK((s_i == cols && s_j == 1) || (s_i == 1 && s_j == rows),
           "mixer: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid", s_i, s_j, rows, cols);
  return FV_OK;
}

}  // namespace

extern "C" int fv_mixer_conv_pool_fwd(const void* xz, const float* conv_w, const float* conv_b,
                                      const float* conv_w_b, const float* conv_b_b, void* xc, int batch,
                                      int rows, int cols, int tok_stride_row, int tok_stride_col, int tokens_per_patch,
                                      int d_inner, int d_conv, int pool_max, float scaling_factor, int dtype,
                                      fv_stream_t stream) {
  int rc = check_geo(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype, tokens_per_patch);
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  FV_CHECK(xz && conv_w && conv_w_b && xc, "mixer_conv_pool_fwd: null pointer");
  FwdParams p{};
  p.xz = xz; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b; p.xc = xc;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner; p.pool_max = pool_max;
  p.pool_scale = pool_max ? 1.f : scaling_factor / (float)cols;
  return dtype == FV_F32 ? dispatch_vec<float>(0, p, (hipStream_t)stream)
                         : dispatch_vec<bf16_t>(0, p, (hipStream_t)stream);
}

extern "C" int fv_mixer_combine_fwd(const void* xz, const float* yc, const float* conv_w, const float* conv_b,
                                    const float* conv_w_b, const float* conv_b_b, const float* D,
                                    const float* D_b, const float* ln_w, const float* ln_b, float ln_eps,
                                    void* g, void* xhat, float* mean, float* rstd, int batch, int rows,
                                    int cols, int tok_stride_row, int tok_stride_col, int tokens_per_patch,
                                    int d_inner, int d_conv, int dtype, fv_stream_t stream) {
  int rc = check_geo(batch, rows, cols, tok_stride_row, tok_stride_col, d_inner, dtype, tokens_per_patch);
  if (rc) return rc;
  FV_CHECK(d_conv == CW, "mixer: only d_conv == %d is built (got %d)", CW, d_conv);
  FV_CHECK(xz && yc && conv_w && conv_w_b && D && D_b && g, "mixer_combine_fwd: null pointer");
  FV_CHECK(!ln_w || (ln_b && mean && rstd), "mixer_combine_fwd: LayerNorm needs weight, bias, mean, rstd");
  FwdParams p{};
  p.xz = xz; p.yc = yc; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b;
  p.Df = D; p.Db = D_b; p.lnw = ln_w; p.lnb = ln_b; p.eps = ln_eps; p.g = g; p.xhat = xhat; p.mean = mean; p.rstd = rstd;
  p.use_norm = ln_w != nullptr;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, tokens_per_patch);
  p.B = batch; p.d_in = d_inner;
  return dtype == FV_F32 ? dispatch_vec<float>(1, p, (hipStream_t)stream)
                         : dispatch_vec<bf16_t>(1, p, (hipStream_t)stream);
}
