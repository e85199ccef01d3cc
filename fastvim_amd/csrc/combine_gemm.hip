// hipcc-flags: -fno-slp-vectorize -fgpu-flush-denormals-to-zero
// combine (expand + skip + average + LayerNorm + SiLU gate) as the A-TILE PRODUCER of the out_proj GEMM whose epilogue is
// the next block's DropPath scale + residual add + RMSNorm -- fv_mixer_combine_fwd + fv_gemm_bf16_addnorm in ONE launch,
// d_inner = 384, d_model = 192, bf16 (FastVim-T).  Reference: mamba_simple_faster.py:356, 412-414, 434-444 and
// models/fastvim.py:168-190.
//
// Why the two fit without a hand-off: both are partitioned by tokens.  A workgroup owns four pooling rows of one image
// (up to 64 tokens: tile row 16 w + j = column j of pooling row i0 + w);
//   phase 1  each of its four waves gates ONE pooling row with combine_fwd_wave's per-token arithmetic (a wave owns a
//            whole token: LayerNorm sums are DPP reductions; one yc row per wave), stores g to HBM once -- backward
//            needs it -- and into a 64 x 384 bf16 A panel in LDS;
//   phase 2  the panel never moves again: the K loop has no barrier and no A traffic; wave w owns output columns
//            [48 w, 48 w + 48) and streams ITS quarter of W_out straight from L2 into MFMA operand registers (a lane's
//            fragment is 16 contiguous bytes of one weight row), six 32-deep k steps ahead in line pairs -- the first six
//            are requested at kernel entry and arrive under phase 1;
//   phase 3  the epilogue of gemm_addnorm_kernel<64>, lane for lane: product rounded to bf16 into an LDS tile, then
//            residual add + RMSNorm of whole 192-wide rows (the residual rows are requested at kernel entry too).
// g is read back by nobody in the forward pass: 19.3 MB of reads and one launch boundary per block go away.
// LDS: 50 KB per workgroup, two workgroups per CU.
#include "mixer_common.h"
#include "packed.h"

namespace {

typedef __bf16 cg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CG_K = 384, CG_N = 192, CG_BM = 64, CG_NT = 256;
constexpr int CG_RSA = CG_K * 2 + 16;          // A panel row stride (bytes): = 4 dwords (mod 64), fragment reads conflict-free
constexpr int CG_RSB = CG_N * 2 + 16;          // product tile row stride of the add + norm epilogue (gemm_mfma.hip)
constexpr int CG_PD = 6;                       // k steps of weight fragments in flight per wave (3: the K loop took 8 us)
constexpr int CG_KS = CG_K / 32;               // 12 k steps
#ifndef CG_DBG
#define CG_DBG 0      // phase probes (tools/probe/r05_combine_phases.sh): 1 no gating loop, 2 no K loop, 3 no epilogue, 4 no g stores
#endif

struct CgParams {
  const void* xz;          // (B, L, 2 * 384) bf16: the z half is read
  const void* skip;        // (B, L, 384) bf16
  const float* yc;         // (2, B, rows, 384) fp32
  const float *lnw, *lnb;  // (384)
  void* g;                 // (B, L, 384) bf16 out (saved for backward)
  float *mean, *rstd_ln;   // (B * L) out
  float ln_eps;
  Geo geo;
  int B;
  const bf16_t* W;         // (192, ldw) bf16, row-major: out_proj.weight
  long ldw;
  const float* residual;   // (M, 192) fp32
  const float* nw;         // (192) RMSNorm weight
  const float* row_scale;  // DropPath scale per sample, nullable
  int rows_per_scale;
  float* res_out;          // (M, 192) fp32
  bf16_t* y;               // (M, 192) normalised rows
  float* rstd;             // (M)
  float eps;
  int M;
  int d_in;                // = 384, as a RUN-TIME value: 1 / d_in must come out of the same v_rcp_f32 as in combine_fwd_wave_kernel
};

__device__ __forceinline__ float cg_hsum(f2 v) { return v.x + v.y; }

__global__ __launch_bounds__(CG_NT, 2) void combine_out_proj_addnorm_kernel(CgParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef PairVec<bf16_t, 3> P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = p.geo;
  // A workgroup owns up to four pooling rows of ONE image (blockIdx.x = image * tiles + tile), a wave one pooling row:
  // tile row 16 w + j is the token at column j of pooling row i0 + w (j < cols <= 16; the other tile rows are zero).  A
  // wave then needs ONE yc row for all its tokens, like the stand-alone kernel -- re-reading it per token (a tile of 64
  // consecutive memory tokens) pulled 77 MB through L2 per launch and held this phase at 20 us.
  const int tiles = (g.rows + 3) / 4;
  const int bimg = blockIdx.x / tiles, i0 = (blockIdx.x - bimg * tiles) * 4;
  const int mimg = bimg * g.L;                       // first memory token of the image
  auto tok_of = [&](int rl) {                        // tile row -> memory token, or -1
    const int w = rl >> 4, j = rl & 15;
    return (i0 + w < g.rows && j < g.cols) ? mimg + (i0 + w) * g.s_i + j * g.s_j : -1;
  };

  // ---- phase 1: gate the wave's pooling row (combine_fwd_wave_kernel<bf16, 3, 1, 2>'s arithmetic, token by token).
  //      Loads return in order, so they are REQUESTED in the order they are needed: the scan outputs and the first two
  //      token pairs, the LayerNorm parameters (no branch around them: an absent norm reads yc and ignores it), then --
  //      used last -- the weight fragments of the first k steps.
  const int lr = lane % 16, gr = lane / 16;
  const int fn = lane & 15, fk = lane >> 4;
  const bf16_t* wrow[3];
  cg_bf16x8 fb[CG_PD][3];
  float4 ne_r[2][2][3], ne_w[3];
  float ne_sc[2][2];
  {
    // (source-order arithmetic, as in combine_fwd_wave_kernel: the two gate tokens bit for bit alike)
#pragma clang fp reassociate(off) contract(off)
    const int lc = lane * 6, voff = lc * 2;
    const int tok_x = 2 * CG_K * 2, tok_s = CG_K * 2;
    const bool has_ln = p.lnw != nullptr;
    const float inv_d = 1.f / (float)p.d_in;      // (a compile-time 1 / 384 is the exactly rounded quotient: one ulp off the stand-alone kernel's)
    const size_t ydir = (size_t)p.B * g.rows * CG_K;
    const __amdgpu_buffer_rsrc_t bz = fv_make_buf((const bf16_t*)p.xz + CG_K, (size_t)p.M * tok_x - tok_s);
    const __amdgpu_buffer_rsrc_t bs = fv_make_buf(p.skip, (size_t)p.M * tok_s);
    const __amdgpu_buffer_rsrc_t bg = fv_make_buf(p.g, (size_t)p.M * tok_s);
    const __amdgpu_buffer_rsrc_t bmean = fv_make_buf(has_ln ? (const void*)p.mean : p.g, has_ln ? (size_t)p.M * 4 : 0);
    const __amdgpu_buffer_rsrc_t brstd = fv_make_buf(has_ln ? (const void*)p.rstd_ln : p.g, has_ln ? (size_t)p.M * 4 : 0);
    const int irow = i0 + wv;
    const bool wlive = irow < g.rows;                  // (a tile's last waves may have no pooling row)
    const int irc = wlive ? irow : g.rows - 1;
    const int mbase = mimg + irc * g.s_i;              // memory token of column 0 of this wave's row
    char* arow = smem + (wv * 16) * CG_RSA + lane * 12;
    constexpr int TT = 2;
    const int npair = (g.cols + TT - 1) / TT;          // <= 8
    f2 ysum[3], ysf[3], ysb[3];
    {
      const float* yr = p.yc + ((size_t)bimg * g.rows + irc) * CG_K + lc;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        ysf[q] = *reinterpret_cast<const f2*>(yr + 2 * q);
        ysb[q] = *reinterpret_cast<const f2*>(yr + ydir + 2 * q);
      }
    }
    // two token pairs are always in flight (requested two pairs ahead)
    struct Pre { P sk[TT], z[TT]; };
    Pre pa, pb;
    auto fetch = [&](Pre& r, int j0) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const int m = mbase + min(j0 + t, g.cols - 1) * g.s_j;
        r.sk[t].load(bs, voff, m * tok_s);
        r.z[t].load(bz, voff, m * tok_x);
      }
    };
    fetch(pa, 0);
    fetch(pb, TT);
    __builtin_amdgcn_sched_barrier(0);
    f2 lw[3], lb[3];
    {
      const float* lwp = has_ln ? p.lnw : p.yc;
      const float* lbp = has_ln ? p.lnb : p.yc;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        lw[q] = *reinterpret_cast<const f2*>(lwp + lc + 2 * q);
        lb[q] = *reinterpret_cast<const f2*>(lbp + lc + 2 * q);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) wrow[nb] = p.W + (long)(wv * 48 + nb * 16 + fn) * p.ldw + fk * 8;
#pragma unroll
    for (int s = 0; s < CG_PD; s += 2)
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) fb[s + h][nb] = *reinterpret_cast<const cg_bf16x8*>(wrow[nb] + (s + h) * 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 3; ++q) ysum[q] = ysf[q] + ysb[q];
    auto process = [&](Pre& r, int j0) {
      f2 o[TT][3], z[TT][3];
      float s1[TT], mean[TT], rstd[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          o[t][q] = r.sk[t].get(q);
          z[t][q] = r.z[t].get(q);
        }
      fetch(r, j0 + 2 * TT);                   // (columns past the row are clamped onto its last token: harmless re-reads)
      __builtin_amdgcn_sched_barrier(0);       // (the refill is requested before this pair's arithmetic, not after it)
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        f2 acc = splat(0.f);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          o[t][q] = (ysum[q] + o[t][q]) * 0.5f;
          acc += o[t][q];
        }
        s1[t] = cg_hsum(acc);
      }
      if (has_ln) {     // mean, then the centred second moment (two exact passes over registers)
#pragma unroll
        for (int t = 0; t < TT; ++t) s1[t] = wave_sum_uniform(s1[t]);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          mean[t] = s1[t] * inv_d;
          f2 acc = splat(0.f);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const f2 d = o[t][q] - mean[t];
            acc = fma2(d, d, acc);
          }
          s1[t] = cg_hsum(acc);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) s1[t] = wave_sum_uniform(s1[t]);
#pragma unroll
        for (int t = 0; t < TT; ++t) rstd[t] = rsqrtf(s1[t] * inv_d + p.ln_eps);
      } else {
#pragma unroll
        for (int t = 0; t < TT; ++t) { mean[t] = 0.f; rstd[t] = 1.f; }
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const bool live = wlive && j0 + t < g.cols;
        const int m = mbase + min(j0 + t, g.cols - 1) * g.s_j;
        f2 out[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f2 w_ = has_ln ? lw[q] : splat(1.f), b_ = has_ln ? lb[q] : splat(0.f);
          out[q] = fma2((o[t][q] - mean[t]) * rstd[t], w_, b_) * silu2(z[t][q]);
        }
        uint32_t pk[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) pk[q] = live ? pack_bf16x2(out[q].x, out[q].y) : 0u;
        // every store is issued unconditionally (a dead token, or a lane other than 0 for the statistics, lands beyond
        // its buffer descriptor and is dropped): a store behind a branch makes the number of memory operations younger
        // than the next pair's loads unknown to the compiler, and it then waits for ALL of them -- the write latency of
        // this pair's g in front of every iteration.  (The range check covers the per-lane offset only: that is where
        // "nowhere" goes; the scalar token offset stays valid.)
        if (CG_DBG != 4) {
          fv_buf_store_words<3>(bg, live ? voff : 0x7ffffff0, m * tok_s, pk);
          const int so = (has_ln && live && lane == 0) ? 0 : 0x7ffffff0;
          const uint32_t mv[1] = {__float_as_uint(mean[t])}, rv[1] = {__float_as_uint(rstd[t])};
          fv_buf_store_words<1>(bmean, so, m * 4, mv);
          fv_buf_store_words<1>(brstd, so, m * 4, rv);
        }
        // A panel row of this token: lane l holds k = 6 l .. 6 l + 5 (three dwords, bank = 3 l + q: conflict-free)
        uint32_t* ar = reinterpret_cast<uint32_t*>(arow + (j0 + t) * CG_RSA);
        ar[0] = pk[0]; ar[1] = pk[1]; ar[2] = pk[2];
      }
    };
#pragma unroll 1
    for (int pr = 0; pr < (CG_DBG == 1 ? 0 : 8); pr += 2) {     // all 16 tile rows are written (dead ones with zeros)
      process(pa, pr * TT);
      process(pb, (pr + 1) * TT);
    }
    (void)npair;
    // the residual rows of the epilogue (lane mapping of add_norm_fwd3_kernel<16>: 16 lanes x 12 channels per row, 4 rows per
    // wave step): requested here, they arrive under the K loop and cost the gating loop no registers
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = tok_of(wv * 16 + it * 8 + u * 4 + gr), rowc = m >= 0 ? m : mimg;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          ne_r[it][u][k] = *reinterpret_cast<const float4*>(p.residual + (size_t)rowc * CG_N + (k * 16 + lr) * 4);
        ne_sc[it][u] = (p.row_scale ? p.row_scale : p.nw)[p.row_scale ? rowc / p.rows_per_scale : 0];
      }
#pragma unroll
    for (int k = 0; k < 3; ++k) ne_w[k] = *reinterpret_cast<const float4*>(p.nw + (k * 16 + lr) * 4);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();       // the panel is complete

  // ---- phase 2: C (64 x 48 per wave) = panel @ W[48 w .. 48 w + 48)^T; no barrier, the weight ring refilled three steps ahead
  cg_f32x4 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (cg_f32x4){0.f, 0.f, 0.f, 0.f};
  const char* afrag = smem + (lane & 15) * CG_RSA + (lane >> 4) * 16;
  // k steps go in PAIRS: a lane's fragments of steps 2 j and 2 j + 1 are the two halves of one 128-byte line of its weight
  // row, and a wave's load instruction touches 16 such rows -- refilled one step at a time, every line was pulled from L2
  // twice (the 32 KB L1 does not keep it for a step: 8 waves x 6 KB pass through in between)
  static_assert(CG_PD % 2 == 0 && CG_KS % 2 == 0, "the weight ring is refilled in line pairs");
#pragma unroll
  for (int kp = 0; kp < (CG_DBG == 2 ? 0 : CG_KS); kp += 2) {
    cg_bf16x8 cur[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) cur[h][nb] = fb[(kp + h) % CG_PD][nb];
    if (kp + CG_PD < CG_KS) {
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          fb[(kp + h) % CG_PD][nb] = *reinterpret_cast<const cg_bf16x8*>(wrow[nb] + (kp + h + CG_PD) * 32);
    }
    // the refill stays HERE: left alone, the scheduler sinks every load to its first use (shorter live ranges) and each
    // k step then waits out a full L2 round trip
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      cg_bf16x8 fa[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) fa[mb] = *reinterpret_cast<const cg_bf16x8*>(afrag + mb * 16 * CG_RSA + (kp + h) * 64);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[h][a], fa[b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();       // every wave is done reading the panel: the product tile overlays it

  // ---- phase 3: residual add + RMSNorm of the tile's rows -- gemm_bf16_body<..., NORM_EPI = 1> of gemm_mfma.hip from the
  //      product tile on, same lane mapping and operation order (bit-identical outputs for the same g)
  //      acc[a][b][j] = C[m = b * 16 + (lane & 15)][n = 48 wv + a * 16 + (lane >> 4) * 4 + j]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const cg_f32x4 v = acc[a][b];
      uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *reinterpret_cast<uint2*>(smem + (b * 16 + (lane & 15)) * CG_RSB + (wv * 48 + a * 16 + (lane >> 4) * 4) * 2) = pk;
    }
  __syncthreads();
  if (CG_DBG == 3) return;
  constexpr int LPR = 16, RPW = 4, RU = 2, RW = CG_BM / 4;
  const float inv_n = 1.f / (float)CG_N;
  float w[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 t = ne_w[k];
    w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
  }
#pragma unroll
  for (int it = 0; it < RW / (RPW * RU); ++it) {
    float v[RU][3][4], r[RU][3][4], sc_u[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int rl = wv * RW + it * (RPW * RU) + u * RPW + gr;       // row of the tile
      sc_u[u] = p.row_scale ? ne_sc[it][u] : 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        const uint2 xb = *reinterpret_cast<const uint2*>(smem + rl * CG_RSB + c * 2);
        v[u][k][0] = __uint_as_float(xb.x << 16); v[u][k][1] = __uint_as_float(xb.x & 0xffff0000u);
        v[u][k][2] = __uint_as_float(xb.y << 16); v[u][k][3] = __uint_as_float(xb.y & 0xffff0000u);
        const float4 t = ne_r[it][u][k];
        r[u][k][0] = t.x; r[u][k][1] = t.y; r[u][k][2] = t.z; r[u][k][3] = t.w;
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = tok_of(wv * RW + it * (RPW * RU) + u * RPW + gr);
      const bool live = row >= 0;
      const size_t base = (size_t)(live ? row : mimg) * CG_N;
      const float sc = sc_u[u];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][k][e] = fmaf(v[u][k][e], sc, r[u][k][e]);
        if (live) *reinterpret_cast<float4*>(p.res_out + base + c) = make_float4(v[u][k][0], v[u][k][1], v[u][k][2], v[u][k][3]);
      }
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) q = fmaf(v[u][k][e], v[u][k][e], q);
#define FV_DPP_ADD(ctrl) q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), ctrl, 0xf, 0xf, true))
      FV_DPP_ADD(0xB1);
      FV_DPP_ADD(0x4E);
      FV_DPP_ADD(0x141);
      FV_DPP_ADD(0x140);
#undef FV_DPP_ADD
      const float rstd = rsqrtf(q * inv_n + p.eps);
      if (lr == 0 && live) p.rstd[row] = rstd;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * LPR + lr) * 4;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = v[u][k][e] * rstd;          // (r * rstd) * w, pinned: the order add_norm_fwd3_kernel uses
          asm volatile("" : "+v"(t));
          o[e] = t * w[k][e];
        }
        uint2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        if (live) *reinterpret_cast<uint2*>(p.y + base + c) = pk;
      }
    }
  }
}

}  // namespace

extern "C" int fv_mixer_combine_out_proj_addnorm_ok(int batch, int rows, int cols, int tokens_per_patch, int d_inner,
                                                    int d_model, int dtype) {
  return dtype == FV_BF16 && d_inner == CG_K && d_model == CG_N && tokens_per_patch == 1 && batch > 0 && rows > 0 &&
         cols > 0 && cols <= 16 && (long)batch * rows * cols * CG_K * 4 < 0x7fffffffL;      // 32-bit byte offsets into xz
}

extern "C" int fv_mixer_combine_out_proj_addnorm(const void* xz, const void* skip, const float* yc, const float* ln_w,
                                                 const float* ln_b, float ln_eps, void* g, float* mean, float* rstd_ln,
                                                 int batch, int rows, int cols, int tok_stride_row, int tok_stride_col,
                                                 const void* W, long ldw, const float* residual,
                                                 const float* norm_weight, const float* row_scale, int rows_per_scale,
                                                 void* y, float* residual_out, float* rstd, float eps,
                                                 fv_stream_t stream) {
  FV_CHECK(xz && skip && yc && g && W && residual && norm_weight && y && residual_out && rstd,
           "mixer_combine_out_proj_addnorm: null pointer");
  FV_CHECK(!ln_w || (ln_b && mean && rstd_ln), "mixer_combine_out_proj_addnorm: LayerNorm needs weight, bias, mean, rstd");
  FV_CHECK(fv_mixer_combine_out_proj_addnorm_ok(batch, rows, cols, 1, CG_K, CG_N, FV_BF16),
           "mixer_combine_out_proj_addnorm: shape not built (fv_mixer_combine_out_proj_addnorm_ok)");
  FV_CHECK((tok_stride_row == cols && tok_stride_col == 1) || (tok_stride_row == 1 && tok_stride_col == rows),
           "mixer_combine_out_proj_addnorm: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid",
           tok_stride_row, tok_stride_col, rows, cols);
  FV_CHECK(((uintptr_t)W & 15) == 0 && ldw % 8 == 0 && ldw >= CG_K, "mixer_combine_out_proj_addnorm: weight rows must be 16-byte aligned");
  FV_CHECK(((uintptr_t)residual & 15) == 0 && ((uintptr_t)residual_out & 15) == 0 && ((uintptr_t)y & 7) == 0 &&
               ((uintptr_t)norm_weight & 15) == 0, "mixer_combine_out_proj_addnorm: row operands must be 16-byte aligned");
  FV_CHECK(!row_scale || rows_per_scale > 0, "mixer_combine_out_proj_addnorm: rows_per_scale must be positive");
  CgParams p{};
  p.xz = xz; p.skip = skip; p.yc = yc; p.lnw = ln_w; p.lnb = ln_b; p.g = g; p.mean = mean; p.rstd_ln = rstd_ln;
  p.ln_eps = ln_eps;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, 1);
  p.B = batch;
  p.W = (const bf16_t*)W; p.ldw = ldw; p.residual = residual; p.nw = norm_weight; p.row_scale = row_scale;
  p.rows_per_scale = rows_per_scale; p.res_out = residual_out; p.y = (bf16_t*)y; p.rstd = rstd; p.eps = eps;
  p.M = batch * rows * cols;
  p.d_in = CG_K;
  const size_t smem = (size_t)CG_BM * CG_RSA;
  hipLaunchKernelGGL(combine_out_proj_addnorm_kernel, dim3(batch * fv_cdiv(rows, 4)), dim3(CG_NT), smem, (hipStream_t)stream, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
