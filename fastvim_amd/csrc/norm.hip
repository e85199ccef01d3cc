// Fused (DropPath-scale) + residual-add + RMSNorm / LayerNorm, forward and backward.
// Replaces the Triton kernels _layer_norm_fwd_1pass_kernel / _layer_norm_bwd_kernel
// (mamba-1p1p1/mamba_ssm/ops/triton/layernorm.py:66-121, 210-304) behind rms_norm_fn /
// layer_norm_fn (layernorm.py:492-512).
//
// One wave64 per row, 4 channels per lane per step (8 B bf16 / 16 B fp32 accesses), the whole row
// kept in registers between the statistics pass and the normalise pass; statistics by DPP wave
// reductions (two exact passes: mean, then centred second moment).  The per-sample DropPath scale
// of the mixer branch (timm DropPath applied in models/fastvim.py:182-190) is folded into the add,
// and the output cast to the mixer's compute dtype is folded into the store.
#include <stdlib.h>

#include "rowwalk.h"

namespace {

constexpr int MAXK_LIMIT = 8;   // up to 8 * 256 = 2048 channels per row (MAXK = ceil(N / 256) is a template parameter)

struct NormParams {
  const void *x, *res, *dy, *dres_out, *r;
  const float *w, *b, *row_scale, *mean_in, *rstd_in;
  void *y, *res_out, *dx, *dres_in;
  float *mean, *rstd, *pw, *pb;
  int x_dt, res_dt, y_dt, ro_dt, dy_dt, dro_dt, r_dt, dx_dt, dri_dt;
  int M, N, rows_per_scale, is_rms;
  float eps;
};

__device__ __forceinline__ void ld4(const void* p, int dt, size_t idx, float (&v)[4]) {
  if (dt == FV_F32) VecIO<float, 4>::load((const float*)p + idx, v);
  else VecIO<bf16_t, 4>::load((const bf16_t*)p + idx, v);
}
__device__ __forceinline__ void st4(void* p, int dt, size_t idx, const float (&v)[4]) {
  if (dt == FV_F32) VecIO<float, 4>::store((float*)p + idx, v);
  else VecIO<bf16_t, 4>::store((bf16_t*)p + idx, v);
}

template <int MAXK>
__global__ __launch_bounds__(256) void add_norm_fwd_kernel(NormParams p) {
  constexpr int RU = MAXK == 1 ? 4 : (MAXK == 2 ? 2 : 1);   // rows in flight per wave: all their loads are issued before any wait
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);   // scalar
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float inv_n = 1.f / (float)p.N;
  for (int row0 = wave * RU; row0 < p.M; row0 += nwaves * RU) {
    float v[RU][MAXK][4], r[RU][MAXK][4];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u;
      const size_t base = (size_t)row * p.N;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (row < p.M && c < p.N) {
          ld4(p.x, p.x_dt, base + c, v[u][k]);
          if (p.res) ld4(p.res, p.res_dt, base + c, r[u][k]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][k][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u;
      if (row >= p.M) break;
      const size_t base = (size_t)row * p.N;
      const float sc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.f;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < p.N) {
          if (p.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][k][e] = fmaf(v[u][k][e], sc, r[u][k][e]);
          } else if (p.row_scale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][k][e] *= sc;
          }
          if (p.res_out) st4(p.res_out, p.ro_dt, base + c, v[u][k]);
#pragma unroll
          for (int e = 0; e < 4; ++e) s += v[u][k][e];
        }
      }
      float mu = 0.f;
      if (!p.is_rms) mu = wave_sum_uniform(s) * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < p.N) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float d = v[u][k][e] - mu;
            q = fmaf(d, d, q);
          }
        }
      }
      const float rstd = rsqrtf(wave_sum_uniform(q) * inv_n + p.eps);
      if (lane == 0) {
        p.rstd[row] = rstd;
        if (!p.is_rms && p.mean) p.mean[row] = mu;
      }
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < p.N) {
          float w[4], o[4];
          VecIO<float, 4>::load(p.w + c, w);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[u][k][e] - mu) * rstd * w[e];
          if (p.b) {
            float bb[4];
            VecIO<float, 4>::load(p.b + c, bb);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += bb[e];
          }
          st4(p.y, p.y_dt, base + c, o);
        }
      }
    }
  }
}

// backward: dr = rstd * (dxhat - mean(dxhat)[LN only] - xhat * mean(dxhat * xhat)) + dres_out
//           dx = dr * row_scale, dres_in = dr;  per-wave-group partials of dw (and db)
template <int MAXK>
__global__ __launch_bounds__(256) void add_norm_bwd_kernel(NormParams p) {
  __shared__ float s_acc[4][MAXK * 256];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);   // scalar
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float inv_n = 1.f / (float)p.N;
  float aw[MAXK][4], ab[MAXK][4];
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) aw[k][e] = ab[k][e] = 0.f;
  constexpr int RU = MAXK == 1 ? 4 : (MAXK == 2 ? 2 : 1);   // rows in flight per wave
  for (int row0 = wave * RU; row0 < p.M; row0 += nwaves * RU) {
    float rr[RU][MAXK][4], dyv[RU][MAXK][4], gg[RU][MAXK][4];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u;
      const size_t base = (size_t)row * p.N;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (row < p.M && c < p.N) {
          ld4(p.r, p.r_dt, base + c, rr[u][k]);
          ld4(p.dy, p.dy_dt, base + c, dyv[u][k]);
          if (p.dres_out) ld4(p.dres_out, p.dro_dt, base + c, gg[u][k]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rr[u][k][e] = dyv[u][k][e] = 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u;
      if (row >= p.M) break;
      const size_t base = (size_t)row * p.N;
      const float rstd = p.rstd_in[row];
      const float mu = p.is_rms ? 0.f : p.mean_in[row];
      float xh[MAXK][4], dxh[MAXK][4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < p.N) {
          float w[4];
          VecIO<float, 4>::load(p.w + c, w);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[k][e] = (rr[u][k][e] - mu) * rstd;
            dxh[k][e] = dyv[u][k][e] * w[e];
            aw[k][e] = fmaf(dyv[u][k][e], xh[k][e], aw[k][e]);
            ab[k][e] += dyv[u][k][e];
            c1 += dxh[k][e];
            c2 = fmaf(dxh[k][e], xh[k][e], c2);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) xh[k][e] = dxh[k][e] = 0.f;
        }
      }
      c2 = wave_sum_uniform(c2) * inv_n;
      c1 = p.is_rms ? 0.f : wave_sum_uniform(c1) * inv_n;
      const float sc = p.row_scale ? p.row_scale[row / p.rows_per_scale] : 1.f;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int c = (k * 64 + lane) * 4;
        if (c < p.N) {
          float dr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) dr[e] = rstd * (dxh[k][e] - c1 - xh[k][e] * c2);
          if (p.dres_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dr[e] += gg[u][k][e];
          }
          if (p.dres_in) st4(p.dres_in, p.dri_dt, base + c, dr);
          if (p.dx) {
            if (p.row_scale)
#pragma unroll
              for (int e = 0; e < 4; ++e) dr[e] *= sc;
            st4(p.dx, p.dx_dt, base + c, dr);
          }
        }
      }
    }
  }
  // block-level fixed-order reduction of the 4 waves' accumulators -> one partial row per block
  for (int pass = 0; pass < (p.pb ? 2 : 1); ++pass) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < p.N)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_acc[wv][c + e] = pass ? ab[k][e] : aw[k][e];
    }
    __syncthreads();
    float* dst = (pass ? p.pb : p.pw) + (size_t)blockIdx.x * p.N;
    for (int c = threadIdx.x; c < p.N; c += blockDim.x)
      dst[c] = (s_acc[0][c] + s_acc[1][c]) + (s_acc[2][c] + s_acc[3][c]);
  }
}

// ---------------------------------------------------------------------------------------------------
// Hidden sizes 192 / 384 / 768 (FastVim-T / S / B): a row is 3 x 4 channels per lane over LPR = 16 / 32 / 64
// lanes, so a wave carries 4 / 2 / 1 rows at full lane occupancy (the generic mapping above, one row per
// wave and 4 channels per lane per step, leaves a quarter of the lanes idle at 192 and 384).  Row sums are
// DPP adds inside a 16-lane row plus the gfx950 cross-row swaps.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#define FV_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
  FV_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
  FV_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
  FV_DPP_ADD(0x141);   // row_half_mirror
  FV_DPP_ADD(0x140);   // row_mirror -> every lane holds its 16-lane sum
#undef FV_DPP_ADD
  if constexpr (LPR >= 32) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if constexpr (LPR >= 64) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}
// sum over the 64 / LPR row groups of a wave (same lane-in-row), result in every group
template <int LPR>
__device__ __forceinline__ float cross_group_sum(float v) {
  if constexpr (LPR <= 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if constexpr (LPR <= 32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void add_norm_fwd3_kernel(NormParams p) {
  constexpr int RPW = 64 / LPR, RU = 2, N = 12 * LPR;   // rows per wave step, row groups in flight
  const int lane = threadIdx.x & 63;
  const int lr = lane % LPR, gr = lane / LPR;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float inv_n = 1.f / (float)N;
  float w[3][4], bb[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    VecIO<float, 4>::load(p.w + (k * LPR + lr) * 4, w[k]);
    if (p.b) VecIO<float, 4>::load(p.b + (k * LPR + lr) * 4, bb[k]);
  }
  for (int row0 = wave * RPW * RU; row0 < p.M; row0 += nwaves * RPW * RU) {
    float v[RU][3][4], r[RU][3][4], sc_u[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * RPW + gr;
      const int rowc = row < p.M ? row : p.M - 1;
      const size_t base = (size_t)rowc * N;
      sc_u[u] = p.row_scale ? p.row_scale[rowc / p.rows_per_scale] : 1.f;    // DropPath scale: rides with the row's loads
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        ld4(p.x, p.x_dt, base + c, v[u][k]);
        if (p.res) ld4(p.res, p.res_dt, base + c, r[u][k]);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * RPW + gr;
      const bool live = row < p.M;
      const size_t base = (size_t)(live ? row : p.M - 1) * N;
      const float sc = sc_u[u];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        if (p.res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][k][e] = fmaf(v[u][k][e], sc, r[u][k][e]);
        } else if (p.row_scale) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][k][e] *= sc;
        }
        if (p.res_out && live) st4(p.res_out, p.ro_dt, base + c, v[u][k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[u][k][e];
      }
      float mu = 0.f;
      if (!p.is_rms) mu = group_sum<LPR>(s) * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[u][k][e] - mu;
          q = fmaf(d, d, q);
        }
      const float rstd = rsqrtf(group_sum<LPR>(q) * inv_n + p.eps);
      if (lr == 0 && live) {
        p.rstd[row] = rstd;
        if (!p.is_rms && p.mean) p.mean[row] = mu;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // ((r - mu) * rstd) * w, in this order whatever -ffast-math would like: fv_gemm_bf16_addnorm's epilogue
          // repeats it and must round the same way
          float t = (v[u][k][e] - mu) * rstd;
          asm volatile("" : "+v"(t));
          o[e] = t * w[k][e] + (p.b ? bb[k][e] : 0.f);
        }
        if (live) st4(p.y, p.y_dt, base + c, o);
      }
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void add_norm_bwd3_kernel(NormParams p) {
  constexpr int RPW = 64 / LPR, RU = 2, N = 12 * LPR;
  __shared__ float s_acc[4][N];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane % LPR, gr = lane / LPR;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float inv_n = 1.f / (float)N;
  float w[3][4], aw[3][4], ab[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    VecIO<float, 4>::load(p.w + (k * LPR + lr) * 4, w[k]);
#pragma unroll
    for (int e = 0; e < 4; ++e) aw[k][e] = ab[k][e] = 0.f;
  }
  for (int row0 = wave * RPW * RU; row0 < p.M; row0 += nwaves * RPW * RU) {
    float rr[RU][3][4], dyv[RU][3][4], gg[RU][3][4];
    float rstd_u[RU], mu_u[RU], sc_u[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * RPW + gr;
      const int rowc = row < p.M ? row : p.M - 1;
      const size_t base = (size_t)rowc * N;
      // the per-row statistics ride with the row's loads: read where they are used, the second row's would wait
      // behind the first row's stores (no aliasing guarantee) -- one more memory round trip in a one-pass kernel
      rstd_u[u] = p.rstd_in[rowc];
      mu_u[u] = p.is_rms ? 0.f : p.mean_in[rowc];
      sc_u[u] = p.row_scale ? p.row_scale[rowc / p.rows_per_scale] : 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        ld4(p.r, p.r_dt, base + c, rr[u][k]);
        ld4(p.dy, p.dy_dt, base + c, dyv[u][k]);
        if (p.dres_out) ld4(p.dres_out, p.dro_dt, base + c, gg[u][k]);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * RPW + gr;
      const bool live = row < p.M;
      const int rowc = live ? row : p.M - 1;
      const size_t base = (size_t)rowc * N;
      const float rstd = rstd_u[u];
      const float mu = mu_u[u];
      const float lv = live ? 1.f : 0.f;                 // a clamped (repeated) row adds nothing
      float xh[3][4], dxh[3][4];
      float c1 = 0.f, c2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dyl = dyv[u][k][e] * lv;
          xh[k][e] = (rr[u][k][e] - mu) * rstd;
          dxh[k][e] = dyl * w[k][e];
          aw[k][e] = fmaf(dyl, xh[k][e], aw[k][e]);
          ab[k][e] += dyl;
          c1 += dxh[k][e];
          c2 = fmaf(dxh[k][e], xh[k][e], c2);
        }
      c2 = group_sum<LPR>(c2) * inv_n;
      c1 = p.is_rms ? 0.f : group_sum<LPR>(c1) * inv_n;
      const float sc = sc_u[u];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        float dr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dr[e] = rstd * (dxh[k][e] - c1 - xh[k][e] * c2) + (p.dres_out ? gg[u][k][e] : 0.f);
        if (p.dres_in && live) st4(p.dres_in, p.dri_dt, base + c, dr);
        if (p.dx && live) {
          if (p.row_scale)
#pragma unroll
            for (int e = 0; e < 4; ++e) dr[e] *= sc;
          st4(p.dx, p.dx_dt, base + c, dr);
        }
      }
    }
  }
  // row groups of the wave first (cross-row swaps), then the 4 waves through LDS, fixed order
  for (int pass = 0; pass < (p.pb ? 2 : 1); ++pass) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = cross_group_sum<LPR>(pass ? ab[k][e] : aw[k][e]);
        if (gr == 0) s_acc[wv][(k * LPR + lr) * 4 + e] = t;
      }
    __syncthreads();
    float* dst = (pass ? p.pb : p.pw) + (size_t)blockIdx.x * N;
    for (int c = threadIdx.x; c < N; c += blockDim.x)
      dst[c] = (s_acc[0][c] + s_acc[1][c]) + (s_acc[2][c] + s_acc[3][c]);
  }
}

int dt_ok(int dt) { return dt == FV_F32 || dt == FV_BF16; }

}  // namespace

extern "C" int fv_add_norm_blocks(int M) {
  static const int cap = fv_tune("FASTVIM_NORM_WAVES", 8192);   // tuning hook (round 3: 8192 from 4096 -- FastVim-B 30.4 -> 30.2 ms,
                                                                // FastVim-T unchanged)
  long waves = M < cap ? M : cap;   // persistent: at most 2048 blocks of 4 waves
  return (int)((waves + 3) / 4);
}

extern "C" int fv_add_norm_fwd(const void* x, int x_dtype, const void* residual, int residual_dtype,
                               const float* weight, const float* bias, const float* row_scale,
                               int rows_per_scale, void* y, int y_dtype, void* residual_out,
                               int residual_out_dtype, float* mean, float* rstd, int M, int N, float eps,
                               int is_rms_norm, fv_stream_t stream) {
  FV_CHECK(M > 0 && N > 0, "add_norm_fwd: empty input");
  FV_CHECK(N % 4 == 0 && N <= MAXK_LIMIT * 256, "add_norm_fwd: hidden size %d must be a multiple of 4 and <= %d", N, MAXK_LIMIT * 256);
  FV_CHECK(x && weight && y && rstd, "add_norm_fwd: null pointer");
  FV_CHECK(dt_ok(x_dtype) && dt_ok(y_dtype) && (!residual || dt_ok(residual_dtype)) &&
               (!residual_out || dt_ok(residual_out_dtype)), "add_norm_fwd: dtypes must be fp32 or bf16");
  FV_CHECK(is_rms_norm || mean, "add_norm_fwd: LayerNorm needs a mean buffer");
  FV_CHECK(!row_scale || rows_per_scale > 0, "add_norm_fwd: rows_per_scale must be positive");
  NormParams p{};
  p.x = x; p.res = residual; p.w = weight; p.b = bias; p.row_scale = row_scale; p.y = y; p.res_out = residual_out;
  p.mean = mean; p.rstd = rstd;
  p.x_dt = x_dtype; p.res_dt = residual_dtype; p.y_dt = y_dtype; p.ro_dt = residual_out_dtype;
  p.M = M; p.N = N; p.rows_per_scale = rows_per_scale; p.is_rms = is_rms_norm; p.eps = eps;
  const dim3 grid(fv_add_norm_blocks(M)), block(256);
  hipStream_t st = (hipStream_t)stream;
  static const bool k3 = (fv_tune("FASTVIM_NORM3", 1) != 0);   // tuning hook
  if (k3 && N == 192) hipLaunchKernelGGL(add_norm_fwd3_kernel<16>, grid, block, 0, st, p);
  else if (k3 && N == 384) hipLaunchKernelGGL(add_norm_fwd3_kernel<32>, grid, block, 0, st, p);
  else if (k3 && N == 768) hipLaunchKernelGGL(add_norm_fwd3_kernel<64>, grid, block, 0, st, p);
  else if (N <= 256) hipLaunchKernelGGL(add_norm_fwd_kernel<1>, grid, block, 0, st, p);
  else if (N <= 512) hipLaunchKernelGGL(add_norm_fwd_kernel<2>, grid, block, 0, st, p);
  else if (N <= 1024) hipLaunchKernelGGL(add_norm_fwd_kernel<4>, grid, block, 0, st, p);
  else hipLaunchKernelGGL(add_norm_fwd_kernel<8>, grid, block, 0, st, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

extern "C" int fv_add_norm_bwd(const void* dy, int dy_dtype, const void* dresidual_out, int dresidual_out_dtype,
                               const void* r, int r_dtype, const float* weight, const float* mean,
                               const float* rstd, const float* row_scale, int rows_per_scale, void* dx,
                               int dx_dtype, void* dresidual_in, int dresidual_in_dtype, float* partial_dw,
                               float* partial_db, int M, int N, int is_rms_norm, fv_stream_t stream) {
  FV_CHECK(M > 0 && N > 0, "add_norm_bwd: empty input");
  FV_CHECK(N % 4 == 0 && N <= MAXK_LIMIT * 256, "add_norm_bwd: hidden size %d must be a multiple of 4 and <= %d", N, MAXK_LIMIT * 256);
  FV_CHECK(dy && r && weight && rstd && partial_dw, "add_norm_bwd: null pointer");
  FV_CHECK(is_rms_norm || mean, "add_norm_bwd: LayerNorm needs the saved mean");
  FV_CHECK(dt_ok(dy_dtype) && dt_ok(r_dtype) && (!dresidual_out || dt_ok(dresidual_out_dtype)) &&
               (!dx || dt_ok(dx_dtype)) && (!dresidual_in || dt_ok(dresidual_in_dtype)),
           "add_norm_bwd: dtypes must be fp32 or bf16");
  NormParams p{};
  p.dy = dy; p.dres_out = dresidual_out; p.r = r; p.w = weight; p.mean_in = mean; p.rstd_in = rstd;
  p.row_scale = row_scale; p.dx = dx; p.dres_in = dresidual_in; p.pw = partial_dw; p.pb = partial_db;
  p.dy_dt = dy_dtype; p.dro_dt = dresidual_out_dtype; p.r_dt = r_dtype; p.dx_dt = dx_dtype; p.dri_dt = dresidual_in_dtype;
  p.M = M; p.N = N; p.rows_per_scale = rows_per_scale; p.is_rms = is_rms_norm;
  const dim3 grid(fv_add_norm_blocks(M)), block(256);
  hipStream_t st = (hipStream_t)stream;
  static const bool k3 = (fv_tune("FASTVIM_NORM3", 1) != 0);   // tuning hook
  if (k3 && N == 192) hipLaunchKernelGGL(add_norm_bwd3_kernel<16>, grid, block, 0, st, p);
  else if (k3 && N == 384) hipLaunchKernelGGL(add_norm_bwd3_kernel<32>, grid, block, 0, st, p);
  else if (k3 && N == 768) hipLaunchKernelGGL(add_norm_bwd3_kernel<64>, grid, block, 0, st, p);
  else if (N <= 256) hipLaunchKernelGGL(add_norm_bwd_kernel<1>, grid, block, 0, st, p);
  else if (N <= 512) hipLaunchKernelGGL(add_norm_bwd_kernel<2>, grid, block, 0, st, p);
  else if (N <= 1024) hipLaunchKernelGGL(add_norm_bwd_kernel<4>, grid, block, 0, st, p);
  else hipLaunchKernelGGL(add_norm_bwd_kernel<8>, grid, block, 0, st, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
