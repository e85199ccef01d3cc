// Backward kernels of the fused FastVim mixer "middle" (channel-last).  Hand-written adjoint of
// csrc/mixer_fwd.hip; replaces the autograd graph of mamba_simple_faster.py:270-444 and the
// hand-written backward of FastVim_MambaInnerFnNoOutProj_withoutZ
// (mamba_ssm/ops/selective_scan_interface.py:607-776):
//
//   fv_mixer_combine_bwd   : d(gate), d(LayerNorm), d(average) -> dz, do, per-row pooled
//                            dyc = 0.5*sum_j do, partials of dLN.weight/bias, dD, dD_b.
//   (fv_mixer_scan_bwd, the adjoint of dt_proj + scan, lives in scan_cl.hip)
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

// out[i] = sum_s in[s*n + i] in a fixed order.  Block = 32 outputs x 8 row-slices: slice q sums rows
// q, q+8, ... (independent 128-B coalesced loads), then the 8 slice sums are added in order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int S, size_t n, int accumulate) {
  __shared__ float s_acc[8][33];
  const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + c;
  float a0 = 0.f, a1 = 0.f;
  if (i < n) {
    int s = q;
    for (; s + 8 < S; s += 16) {
      a0 += in[(size_t)s * n + i];
      a1 += in[(size_t)(s + 8) * n + i];
    }
    if (s < S) a0 += in[(size_t)s * n + i];
  }
  s_acc[q][c] = a0 + a1;
  __syncthreads();
  if (q == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_acc[k][c];
    out[i] = accumulate ? out[i] + t : t;
  }
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

extern "C" int fv_reduce_partials(const float* partials, float* out, int n_partials, size_t n,
                                  fv_stream_t stream) {
  FV_CHECK(partials && out && n_partials > 0, "reduce_partials: bad arguments");
  if (n == 0) return FV_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(fv_cdiv((long)n, 32)), dim3(256), 0, (hipStream_t)stream,
                     partials, out, n_partials, n, 0);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
