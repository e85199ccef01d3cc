// hipcc-flags: -fno-slp-vectorize -fgpu-flush-denormals-to-zero
// Wave-per-token combine kernels (expand + skip + average + LayerNorm + gate, and the adjoint) for d_inner = 384 or 768
// (mamba_simple_faster.py:356, 412-414, 434-441; channel path mamba_simple_channel_faster.py:333-340, 452-478).
//
// One wave owns a whole token: a lane holds NCK chunks of NP channel pairs (chunk k = channels [k*128*NP, (k+1)*128*NP)),
// so the LayerNorm sums are DPP wave reductions only -- no LDS, no block barrier (the generic kernels in
// mixer_fwd.hip / mixer_bwd.hip split a 768-wide token over two waves and pay four block barriers per token pair).
// The unit of work is a pooling GROUP: (batch, row, channel slot) -- its tokens are pcols memory tokens at a fixed
// stride, its scan output is one yc row, its pooled gradient one dyc row, so tokens_per_patch > 1 needs no modulo,
// no LDS slot accumulators and no separate code path.  Arithmetic is explicit 2-wide packed fp32, token access
// is through buffer descriptors (scalar byte offsets), the next TT tokens are in flight while TT are processed.
#include "mixer_common.h"
#include "packed.h"

namespace {

using fvi::BwdParams;
using fvi::FwdParams;

constexpr int NW = 4;     // waves (= groups in flight) per block

__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }

template <typename T, int NP, int NCK, int TT>
__global__ __launch_bounds__(64 * NW) void combine_fwd_wave_kernel(FwdParams p) {
  // source-order arithmetic (no reassociation, no contraction beyond the explicit fmas): csrc/combine_gemm.hip gates tokens
  // with the same expressions inside the out_proj launch, and the two must agree bit for bit whatever surrounds them
#pragma clang fp reassociate(off) contract(off)
  typedef PairVec<T, NP> P;
  constexpr int CHK = 128 * NP, CHKB = CHK * (int)sizeof(T);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Geo g = p.geo;
  const int tpp = g.tpp, pcols = g.pcols;
  const int lc = lane * 2 * NP, voff = lc * (int)sizeof(T);
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_s = p.d_in * (int)sizeof(T);
  f2 lw[NCK][NP], lb[NCK][NP];
#pragma unroll
  for (int k = 0; k < NCK; ++k)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      lw[k][q] = p.use_norm ? load_f2(p.lnw, k * CHK + lc + 2 * q) : splat(1.f);
      lb[k][q] = p.use_norm ? load_f2(p.lnb, k * CHK + lc + 2 * q) : splat(0.f);
    }
  const float inv_d = 1.f / (float)p.d_in;
  const size_t ydir = (size_t)p.B * g.rows * tpp * p.d_in;
  const int ngroups = p.B * g.rows * tpp;
  for (int grp = blockIdx.x * NW + wv; grp < ngroups; grp += gridDim.x * NW) {
    const int row = grp / tpp, c = grp - row * tpp, b = row / g.rows, i = row - b * g.rows;
    const __amdgpu_buffer_rsrc_t bz = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in + p.d_in, (size_t)g.L * tok_x - tok_s);
    const __amdgpu_buffer_rsrc_t bs = fv_make_buf((const T*)p.skip + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_s);
    const __amdgpu_buffer_rsrc_t bg = fv_make_buf((T*)p.g + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_s);
    const float* yr = p.yc + (size_t)grp * p.d_in + lc;
    f2 ys[NCK][NP];
#pragma unroll
    for (int k = 0; k < NCK; ++k)
#pragma unroll
      for (int q = 0; q < NP; ++q) ys[k][q] = load_f2(yr, k * CHK + 2 * q) + load_f2(yr + ydir, k * CHK + 2 * q);
    const int base = i * g.s_i * tpp + c, step = g.s_j * tpp;      // memory token of column j: base + j * step
    P nsk[TT][NCK], nz[TT][NCK];
    auto fetch = [&](int j0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int m = base + (j0 + t) * step;
#pragma unroll
        for (int k = 0; k < NCK; ++k) {
          nsk[t][k].load(bs, voff, m * tok_s + k * CHKB);
          nz[t][k].load(bz, voff, m * tok_x + k * CHKB);
        }
      }
    };
    fetch(0);
    for (int j0 = 0; j0 < pcols; j0 += TT) {       // pcols % TT == 0
      f2 o[TT][NCK][NP], z[TT][NCK][NP];
      float s1[TT], mean[TT], rstd[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int k = 0; k < NCK; ++k)
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            o[t][k][q] = nsk[t][k].get(q);
            z[t][k][q] = nz[t][k].get(q);
          }
      if (j0 + TT < pcols) fetch(j0 + TT);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        f2 acc = splat(0.f);
#pragma unroll
        for (int k = 0; k < NCK; ++k)
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            o[t][k][q] = (ys[k][q] + o[t][k][q]) * 0.5f;
            acc += o[t][k][q];
          }
        s1[t] = hsum(acc);
      }
      if (p.use_norm) {     // mean, then the centred second moment (two exact passes over registers)
#pragma unroll
        for (int t = 0; t < TT; ++t) s1[t] = wave_sum_uniform(s1[t]);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          mean[t] = s1[t] * inv_d;
          f2 acc = splat(0.f);
#pragma unroll
          for (int k = 0; k < NCK; ++k)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              const f2 d = o[t][k][q] - mean[t];
              acc = fma2(d, d, acc);
            }
          s1[t] = hsum(acc);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) s1[t] = wave_sum_uniform(s1[t]);
#pragma unroll
        for (int t = 0; t < TT; ++t) rstd[t] = rsqrtf(s1[t] * inv_d + p.eps);
      } else {
#pragma unroll
        for (int t = 0; t < TT; ++t) { mean[t] = 0.f; rstd[t] = 1.f; }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int m = base + (j0 + t) * step;
#pragma unroll
        for (int k = 0; k < NCK; ++k) {
          f2 out[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q)
            out[q] = fma2((o[t][k][q] - mean[t]) * rstd[t], lw[k][q], lb[k][q]) * silu2(z[t][k][q]);
          P::store(bg, voff, m * tok_s + k * CHKB, out);
        }
        if (p.use_norm && lane == 0) {
          p.mean[(size_t)b * g.L + m] = mean[t];
          p.rstd[(size_t)b * g.L + m] = rstd[t];
        }
      }
    }
  }
}

template <typename T, int NP, int NCK, int TT>
__global__ __launch_bounds__(64 * NW) void combine_bwd_wave_kernel(BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 * d_in
  typedef PairVec<T, NP> P;
  constexpr int CHK = 128 * NP, CHKB = CHK * (int)sizeof(T);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Geo g = p.geo;
  const int tpp = g.tpp, pcols = g.pcols;
  const int lc = lane * 2 * NP, voff = lc * (int)sizeof(T);
  const int tok_x = 2 * p.d_in * (int)sizeof(T), tok_s = p.d_in * (int)sizeof(T);
  f2 lw[NCK][NP], lb[NCK][NP], a_lw[NCK][NP], a_lb[NCK][NP];
#pragma unroll
  for (int k = 0; k < NCK; ++k)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      lw[k][q] = p.use_norm ? load_f2(p.lnw, k * CHK + lc + 2 * q) : splat(1.f);
      lb[k][q] = p.use_norm ? load_f2(p.lnb, k * CHK + lc + 2 * q) : splat(0.f);
      a_lw[k][q] = a_lb[k][q] = splat(0.f);
    }
  const float inv_d = 1.f / (float)p.d_in;
  const size_t ydir = (size_t)p.B * g.rows * tpp * p.d_in;
  const int ngroups = p.B * g.rows * tpp;
  for (int grp = blockIdx.x * NW + wv; grp < ngroups; grp += gridDim.x * NW) {
    const int row = grp / tpp, c = grp - row * tpp, b = row / g.rows, i = row - b * g.rows;
    const __amdgpu_buffer_rsrc_t bdg = fv_make_buf((const T*)p.dg + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_s);
    const __amdgpu_buffer_rsrc_t bz = fv_make_buf((const T*)p.xz + (size_t)b * g.L * 2 * p.d_in + p.d_in, (size_t)g.L * tok_x - tok_s);
    const __amdgpu_buffer_rsrc_t bs = fv_make_buf((const T*)p.skip + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_s);
    const __amdgpu_buffer_rsrc_t bdz = fv_make_buf((T*)p.dxz + (size_t)b * g.L * 2 * p.d_in + p.d_in, (size_t)g.L * tok_x - tok_s);
    const __amdgpu_buffer_rsrc_t bdo = fv_make_buf((T*)p.dob + (size_t)b * g.L * p.d_in, (size_t)g.L * tok_s);
    const float* yr = p.yc + (size_t)grp * p.d_in + lc;
    f2 ys[NCK][NP], dy[NCK][NP];
#pragma unroll
    for (int k = 0; k < NCK; ++k)
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        ys[k][q] = load_f2(yr, k * CHK + 2 * q) + load_f2(yr + ydir, k * CHK + 2 * q);
        dy[k][q] = splat(0.f);
      }
    const int base = i * g.s_i * tpp + c, step = g.s_j * tpp;
    P ndg[TT][NCK], nz[TT][NCK], nsk[TT][NCK];
    auto fetch = [&](int j0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int m = base + (j0 + t) * step;
#pragma unroll
        for (int k = 0; k < NCK; ++k) {
          ndg[t][k].load(bdg, voff, m * tok_s + k * CHKB);
          nz[t][k].load(bz, voff, m * tok_x + k * CHKB);
          nsk[t][k].load(bs, voff, m * tok_s + k * CHKB);
        }
      }
    };
    fetch(0);
    // LayerNorm statistics of the group's tokens: lane l holds token j0 + l of the current 64-token chunk, loaded
    // with the group's first fetch and handed out by v_readlane -- a per-token load inside the loop is a memory
    // round trip in front of every iteration's arithmetic
    float mu_l = 0.f, rs_l = 1.f;
    for (int j0 = 0; j0 < pcols; j0 += TT) {       // pcols % TT == 0
      if (p.use_norm && (j0 & 63) == 0 && j0 + lane < pcols) {
        const size_t sm = (size_t)b * g.L + base + (j0 + lane) * step;
        mu_l = p.mean[sm];
        rs_l = p.rstd[sm];
      }
      f2 xh[TT][NCK][NP], dxh[TT][NCK][NP];
      float c1[TT], c2[TT], rs[TT];
      {
        f2 dg[TT][NCK][NP], z[TT][NCK][NP];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int k = 0; k < NCK; ++k)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              dg[t][k][q] = ndg[t][k].get(q);
              z[t][k][q] = nz[t][k].get(q);
              xh[t][k][q] = nsk[t][k].get(q);
            }
        if (j0 + TT < pcols) fetch(j0 + TT);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const int m = base + (j0 + t) * step;
          rs[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs_l), (j0 + t) & 63));
          const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu_l), (j0 + t) & 63));
          const float hr = 0.5f * rs[t], mr = -mu * rs[t];
          f2 s1 = splat(0.f), s2 = splat(0.f);
#pragma unroll
          for (int k = 0; k < NCK; ++k) {
            f2 dz[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              const f2 x = fma2(ys[k][q] + xh[t][k][q], splat(hr), splat(mr));     // skip -> xhat
              const f2 h = fma2(x, lw[k][q], lb[k][q]);
              const f2 zz = z[t][k][q], sg = sigmoid2(zz), zs = zz * sg;
              const f2 dh = dg[t][k][q] * zs;
              dz[q] = dg[t][k][q] * h * fma2(zs, 1.f - sg, sg);
              a_lw[k][q] = fma2(dh, x, a_lw[k][q]);
              a_lb[k][q] += dh;
              const f2 dx = dh * lw[k][q];
              s1 += dx;
              s2 = fma2(dx, x, s2);
              xh[t][k][q] = x;
              dxh[t][k][q] = dx;
            }
            P::store(bdz, voff, m * tok_x + k * CHKB, dz);
          }
          c1[t] = hsum(s1);
          c2[t] = hsum(s2);
        }
      }
      if (p.use_norm) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          c1[t] = wave_sum_uniform(c1[t]) * inv_d;
          c2[t] = wave_sum_uniform(c2[t]) * inv_d;
        }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int m = base + (j0 + t) * step;
#pragma unroll
        for (int k = 0; k < NCK; ++k) {
          f2 d_o[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            d_o[q] = p.use_norm ? (dxh[t][k][q] - fma2(xh[t][k][q], splat(c2[t]), splat(c1[t]))) * rs[t] : dxh[t][k][q];
            dy[k][q] = fma2(d_o[q], splat(0.5f), dy[k][q]);
          }
          P::store(bdo, voff, m * tok_s + k * CHKB, d_o);
        }
      }
    }
    float* dyr = p.dyc + (size_t)grp * p.d_in + lc;
#pragma unroll
    for (int k = 0; k < NCK; ++k)
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<f2*>(dyr + k * CHK + 2 * q) = dy[k][q];
  }
  // fixed-order accumulation of the block's waves into one partial row [d ln_w | d ln_b]
  for (int w = 0; w < NW; ++w) {
    __syncthreads();
    if (w == wv) {
#pragma unroll
      for (int k = 0; k < NCK; ++k)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int ch = k * CHK + lc + 2 * q + v;
            smem[ch] = (w == 0 ? 0.f : smem[ch]) + a_lw[k][q][v];
            smem[p.d_in + ch] = (w == 0 ? 0.f : smem[p.d_in + ch]) + a_lb[k][q][v];
          }
    }
  }
  __syncthreads();
  float* dst = p.part + (size_t)blockIdx.x * 2 * p.d_in;
  for (int e = threadIdx.x; e < 2 * p.d_in; e += blockDim.x) dst[e] = smem[e];
}

int mode() {   // tuning hook: 0 = generic kernels only, 1 = wave kernels where a token spans more than one 384-chunk, 2 = wherever they apply
  static const int m = fv_tune("FASTVIM_COMBINE_WAVE", 2);
  return m;
}

int chunks(int d_in) {
  if (d_in % 384 != 0 || d_in / 384 > 2) return 0;
  const int nck = d_in / 384;
  return (mode() >= 2 || (mode() == 1 && nck > 1)) ? nck : 0;
}

}  // namespace

static bool wide_bwd(int d_in) {
  static const bool on = (fv_tune("FASTVIM_COMBINE_WAVE_B", 1) != 0);   // tuning hook
  return on && d_in == 4 * 384 && mode() >= 1;
}

int fvi::combine_wave_blocks(int B, int rows, int tpp, int d_in) {
  if (!chunks(d_in) && !wide_bwd(d_in)) return 0;
  const long groups = ((long)B * rows * tpp + NW - 1) / NW;
  static const int cap = fv_tune("FASTVIM_COMBINE_GRID", 0);   // tuning hook
  if (cap) return (int)(groups < cap ? groups : cap);
  const long per = (groups + 511) / 512;
  return (int)((groups + per - 1) / per);
}

#define FV_WAVE_LAUNCH(KERNEL, PARAMS, SMEM)                                                        \
  do {                                                                                             \
    const int nck = chunks(p.d_in);                                                                \
    if (!nck) return FV_ERR_UNSUPPORTED;                                                           \
    FV_CHECK((size_t)p.geo.L * 2 * p.d_in * 4 <= 0xfffff000ull,                                    \
             "mixer_combine: %d tokens x %d channels exceed the 32-bit buffer offsets of the wave kernels", p.geo.L, p.d_in); \
    const dim3 grid(fvi::combine_wave_blocks(p.B, p.geo.rows, p.geo.tpp, p.d_in)), block(64 * NW); \
    const bool two = p.geo.pcols % 2 == 0;                                                         \
    if (dtype == FV_F32) {                                                                         \
      if (nck == 1 && two) hipLaunchKernelGGL((KERNEL<float, 3, 1, 2>), grid, block, SMEM, st, p); \
      else if (nck == 1) hipLaunchKernelGGL((KERNEL<float, 3, 1, 1>), grid, block, SMEM, st, p);   \
      else if (two) hipLaunchKernelGGL((KERNEL<float, 3, 2, 2>), grid, block, SMEM, st, p);        \
      else hipLaunchKernelGGL((KERNEL<float, 3, 2, 1>), grid, block, SMEM, st, p);                 \
    } else {                                                                                       \
      if (nck == 1 && two) hipLaunchKernelGGL((KERNEL<bf16_t, 3, 1, 2>), grid, block, SMEM, st, p);\
      else if (nck == 1) hipLaunchKernelGGL((KERNEL<bf16_t, 3, 1, 1>), grid, block, SMEM, st, p);  \
      else if (two) hipLaunchKernelGGL((KERNEL<bf16_t, 3, 2, 2>), grid, block, SMEM, st, p);       \
      else hipLaunchKernelGGL((KERNEL<bf16_t, 3, 2, 1>), grid, block, SMEM, st, p);                \
    }                                                                                              \
    FV_LAUNCH_CHECK();                                                                             \
    return FV_OK;                                                                                  \
  } while (0)

int fvi::combine_fwd_wave(const FwdParams& p, int dtype, hipStream_t st) {
  // d_inner = 1536 (FastVim-B): four chunks per lane, one token in flight (the backward fits 255 VGPRs exactly)
  if (p.d_in == 4 * 384 && mode() >= 1 && (size_t)p.geo.L * 2 * p.d_in * 4 <= 0xfffff000ull) {
    const long groups = ((long)p.B * p.geo.rows * p.geo.tpp + NW - 1) / NW;
    const long per = (groups + 511) / 512;
    const dim3 grid((int)((groups + per - 1) / per)), block(64 * NW);
    if (dtype == FV_F32) hipLaunchKernelGGL((combine_fwd_wave_kernel<float, 3, 4, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((combine_fwd_wave_kernel<bf16_t, 3, 4, 1>), grid, block, 0, st, p);
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  FV_WAVE_LAUNCH(combine_fwd_wave_kernel, p, 0);
}

int fvi::combine_bwd_wave(const BwdParams& p, int dtype, hipStream_t st) {
  if (wide_bwd(p.d_in) && (size_t)p.geo.L * 2 * p.d_in * 4 <= 0xfffff000ull) {
    const dim3 grid(fvi::combine_wave_blocks(p.B, p.geo.rows, p.geo.tpp, p.d_in)), block(64 * NW);
    const size_t smem = (size_t)2 * p.d_in * 4;
    if (dtype == FV_F32) hipLaunchKernelGGL((combine_bwd_wave_kernel<float, 3, 4, 1>), grid, block, smem, st, p);
    else hipLaunchKernelGGL((combine_bwd_wave_kernel<bf16_t, 3, 4, 1>), grid, block, smem, st, p);
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  FV_WAVE_LAUNCH(combine_bwd_wave_kernel, p, (size_t)2 * p.d_in * 4);
}
