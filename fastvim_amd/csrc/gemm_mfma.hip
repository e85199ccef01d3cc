// bf16 MFMA GEMM for the projections of the FastVim block (in_proj, out_proj, patch-embed, and their
// data- and weight-gradient forms).  Replaces the cuBLAS calls behind F.linear / matmul in
// mamba_simple_faster.py:189-193, 435-444 and models/fastvim.py:95 (Conv2d k=s=16 == a GEMM).
//
//   C[M][N] = sum_k A(m,k) * B(k,n),   bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Operand storage (per operand, template parameter):
//   KC (K-contiguous): A(m,k) = A[m*lda + k];  B(k,n) = B[n*ldb + k]   ("NT": activations x weight^T)
//   KS (K-slow)      : A(m,k) = A[k*lda + m];  B(k,n) = B[k*ldb + n]
// so forward / data-gradient GEMMs are <KC,KC> or <KC,KS> and weight gradients X^T Y are <KS,KS> with
// the reduction over tokens split across grid.z (fp32 partials, summed by fv_reduce_partials).
//
// gfx950 structure: 256 threads = 4 wave64, each wave owns a 64x64 accumulator (4x4 MFMA tiles);
// K tiles of 64 are staged global -> registers -> LDS (double buffered: the loads of tile t+1 are in
// flight while tile t is multiplied).  KC tiles are stored [row][k] with a 16-byte XOR swizzle and read
// with ds_read_b128; KS tiles are stored as loaded, [k][col], and read with the CDNA4 transposing
// LDS read ds_read_b64_tr_b16, so no operand is ever transposed in HBM.  The MFMA is issued with the
// roles swapped (weight-side operand in the A slot), which leaves each lane with 4 consecutive output
// columns of one row: the epilogue stores 8-byte (bf16) / 16-byte (fp32) vectors.
#include <stdlib.h>
#include <utility>

#include "common.h"
#include "gemm_tiles.h"

namespace {


#ifdef FASTVIM_TUNING_HOOKS
// Phase stamps of the fused projection + norm kernels (diagnostic build only; guide section 7 "In-kernel stamps"): wave 0 of
// a workgroup stores s_memtime at its phase boundaries into a buffer nothing else reads.  Set with fv_debug_set_stamps().
unsigned long long* g_fv_stamps = nullptr;
__device__ __forceinline__ void fv_stamp(unsigned long long* buf, int slot) {
  if (buf && threadIdx.x == 0) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    buf[(size_t)blockIdx.x * 8 + slot] = t;
  }
}
#define FV_STAMP(ne, slot) fv_stamp((ne)->stamps, slot)
#else
#define FV_STAMP(ne, slot) ((void)0)
#endif

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;     // (N) fp32, nullable, added in the epilogue; with rb_period > 0 a (rb_period, N) table instead
  int M, N, K;
  int rb_period;         // > 0: fp32 C only, C[m][n] = bf16_round(product) + bias[(m mod rb_period) * N + n]
  long lda, ldb, ldc;
  int k_per_split;       // K range of one grid.z slice (multiple of BK)
  int c_fp32;            // 1: C is fp32, else bf16
  long c_split_stride;   // elements between split-K partials
};
static_assert(sizeof(GemmParams) == 88, "40 of these plus the prefix table must fit the 4 KiB kernel-argument limit");

// Epilogue of the out_proj GEMM fused with the NEXT block's residual add + RMSNorm (fv_gemm_bf16_addnorm): one tile is
// BN = N = 192 wide, so a workgroup owns whole rows.
struct NormEpi {
  const float* residual;   // (M, N) fp32
  const float* w;          // (N) RMSNorm weight
  const float* row_scale;  // per-sample DropPath scale of the GEMM output, nullable
  float* res_out;          // (M, N) fp32: residual + scale * bf16_round(product)
  bf16_t* y;               // (M, N) normalised rows
  float* rstd;             // (M)
  int rows_per_scale;
  float eps;
  // backward form (fv_gemm_bf16_dgrad_addnorm_bwd): the GEMM output is d y of the norm; `residual` is the saved
  // normalisation input r, `rstd` is read, `y` receives d x (bf16, x row_scale) and `res_out` d residual_in
  const float* dres_out;   // (M, N) fp32 gradient of the residual stream arriving from above, nullable
  float* pw;               // (workgroups, N) per-workgroup partial sums of d norm weight
  // second GEMM phase of the backward form, nullable: C2 (M, N2) bf16 = d x (the tile just produced, still in LDS) @ W2
  // (N, N2) as stored -- the previous block's out_proj data gradient
  const bf16_t* W2;
  bf16_t* C2;
  long ldw2;
  int N2;
  // rows of the output a workgroup owns (<= the tile height BM; 0 = BM): trimmed so that every resident workgroup gets the
  // same share of the rows -- M = 25 088 over 512 slots is 49 rows each instead of 392 tiles of 64 (1.53 per slot)
  int rpt;
#ifdef FASTVIM_TUNING_HOOKS
  unsigned long long* stamps;      // (workgroups, 8) phase stamps, nullable
#endif
};

// WM x WN waves, each 16*MB rows x 16*NB columns: tile BM = 16*MB*WM rows (m), BN = 16*NB*WN cols (n).
// NB = 6 gives 128x192 tiles: an N = 192 output is one tile wide, so the A panel is read once instead of 3 times.
// MB = 8 (128-row wave tiles, 256-row block tiles) is for the compute-bound FastVim-S/B widths: a k-step of a
// 64x64 wave tile reads 8 KiB of fragments for 16 MFMAs -- at the full MFMA rate that is exactly the 128 B/clk the
// LDS delivers -- while a 128x64 wave tile reads 12 KiB for 32.
template <int AMODE, int BMODE, int WM, int WN, bool GLDS, int NB = 4, int MB = 4, bool XREMAP = true, int NORM_EPI = 0>
__device__ __forceinline__ void gemm_bf16_body(const GemmParams& p, int block_id, int split, const NormEpi* ne = nullptr,
                                               int dbg = 0) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN, WNC = 16 * NB, WMR = 16 * MB;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto sA = [&](int i) { return smem + i * (A_BYTES + B_BYTES); };             // [A0 | B0 | A1 | B1]
  auto sB = [&](int i) { return smem + i * (A_BYTES + B_BYTES) + A_BYTES; };
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WN, wn = wv % WN;
  // XCD-aware tile order: the N tiles that share an A row panel are consecutive on one XCD
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int nblk = NORM_EPI != 0 ? (int)gridDim.x : tiles_m * tiles_n;      // (trimmed row tiles: the launch knows how many)
  int bid = block_id;
  if constexpr (XREMAP) {      // block_id is the hardware block index (XCD = block_id % 8); the grouped kernel remaps itself
    const int q8 = nblk / 8, r8 = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;   // bijective remap
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int rpt = (NORM_EPI != 0 && ne->rpt > 0) ? ne->rpt : BM;
  const int m0 = tm * rpt, n0 = tn * BN;
  const int mlim = NORM_EPI != 0 ? min(p.M, m0 + rpt) : p.M;      // rows [m0, mlim) are this workgroup's
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;

  if constexpr (NORM_EPI != 0) FV_STAMP(ne, 0);
  f32x4 acc[NB][MB];   // [n tile][m tile]: rows = n (MFMA A slot = B operand), cols = m
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // add + norm epilogue: the residual rows this lane will add are requested now and arrive under the K loop
  constexpr bool NE_PRE = NORM_EPI != 0 && BM <= 64;     // taller tiles: no registers left, the epilogue loads its rows itself
  constexpr int NE_IT = NE_PRE ? BM / 4 / 8 : 1;
  float4 ne_r[NE_IT][2][3];
  float4 ne_g[NORM_EPI == 2 ? NE_IT : 1][2][3];      // backward: the residual-stream gradient rows as well
  if constexpr (NORM_EPI == 2 && NE_PRE) {
    const int lr = lane % 16, gr = lane / 16;
#pragma unroll
    for (int it = 0; it < NE_IT; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = m0 + wv * (BM / 4) + it * 8 + u * 4 + gr, rowc = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          ne_g[it][u][k] = ne->dres_out ? *reinterpret_cast<const float4*>(ne->dres_out + (size_t)rowc * BN + (k * 16 + lr) * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  if constexpr (NE_PRE) {
    const int lr = lane % 16, gr = lane / 16;
#pragma unroll
    for (int it = 0; it < NE_IT; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = m0 + wv * (BM / 4) + it * 8 + u * 4 + gr, rowc = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          ne_r[it][u][k] = *reinterpret_cast<const float4*>(ne->residual + (size_t)rowc * BN + (k * 16 + lr) * 4);
      }
  }

  constexpr bool OPA = GLDS && AMODE == KS, OPB = GLDS && BMODE == KS;
  KsFrags<BM, OPA ? MB : 1> ka;
  KsFrags<BN, OPB ? NB : 1> kb;
  if constexpr (OPA) ka.init(sA(0), wm * MB, lane);
  if constexpr (OPB) kb.init(sB(0), wn * NB, lane);
  auto compute = [&](int cur) {
    // LDS-DMA staging with a K-slow operand: opaque transposing reads (see KsFrags)
    if constexpr (OPA) ka.read(cur * (A_BYTES + B_BYTES));
    if constexpr (OPB) kb.read(cur * (A_BYTES + B_BYTES));
    if constexpr (OPA) ka.wait();
    else if constexpr (OPB) kb.wait();
    if constexpr (OPA && OPB) kb.pin();
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 fa[MB], fb[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        if constexpr (OPA) fa[i] = ka.get(ks, i);
        else fa[i] = frag<AMODE, BM>(sA(cur), wm * MB + i, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if constexpr (OPB) fb[i] = kb.get(ks, i);
        else fb[i] = frag<BMODE, BN>(sB(cur), wn * NB + i, ks, lane);
      }
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[a], fa[b], acc[a][b], 0, 0, 0);
    }
  };
  if constexpr (GLDS) {
    // LDS-DMA: the next tile streams into the other buffer while this one is multiplied; __syncthreads() is
    // exactly the wait it needs (vmcnt(0): my pieces have landed; barrier: everyone's have, and everyone is
    // done reading the buffer that is overwritten next).
    GldsPlan<AMODE, BM, NT> ga;
    GldsPlan<BMODE, BN, NT> gb;
    ga.init(p.A, p.lda, m0, p.M, tid);
    gb.init(p.B, p.ldb, n0, p.N, tid);
    ga.issue(sA(0), kbeg, tid);
    gb.issue(sB(0), kbeg, tid);
    __syncthreads();
    if constexpr (NORM_EPI != 0) FV_STAMP(ne, 1);      // first stage landed
    for (int t = 0; t < nt; ++t) {
      const int cur = t & 1;
#ifdef FASTVIM_TUNING_HOOKS      // phase probe: dbg 1 = no multiply, 2 = no loads, 3 = loads of one K tile only (cache-resident)
      if (t + 1 < nt && dbg != 2) {
        ga.issue(sA(cur ^ 1), kbeg + (dbg == 3 ? 0 : (t + 1) * BK), tid);
        gb.issue(sB(cur ^ 1), kbeg + (dbg == 3 ? 0 : (t + 1) * BK), tid);
      }
      if (dbg != 1) compute(cur);
#else
      if (t + 1 < nt) {
        ga.issue(sA(cur ^ 1), kbeg + (t + 1) * BK, tid);
        gb.issue(sB(cur ^ 1), kbeg + (t + 1) * BK, tid);
      }
      compute(cur);
#endif
      __syncthreads();
    }
  } else {
  Stage<AMODE, BM, NT> ra;
  Stage<BMODE, BN, NT> rb;
  ra.load(p.A, p.lda, m0, kbeg, p.M, kend, tid);
  rb.load(p.B, p.ldb, n0, kbeg, p.N, kend, tid);
  ra.store(sA(0), tid);
  rb.store(sB(0), tid);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      ra.load(p.A, p.lda, m0, kbeg + (t + 1) * BK, p.M, kend, tid);
      rb.load(p.B, p.ldb, n0, kbeg + (t + 1) * BK, p.N, kend, tid);
    }
    compute(cur);
    if (t + 1 < nt) {
      ra.store(sA(cur ^ 1), tid);
      rb.store(sB(cur ^ 1), tid);
    }
    __syncthreads();
  }
  }
  if constexpr (NORM_EPI != 0) FV_STAMP(ne, 2);        // K loop done
  // epilogue: acc[a][b][j] = C[m = m0 + wm*WMR + b*16 + (lane&15)][n = n0 + wn*WNC + a*16 + (lane>>4)*4 + j]
  const long zoff = (long)split * p.c_split_stride;
  if (p.bias && p.rb_period <= 0) {
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      const int n = n0 + wn * WNC + a * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bj = (n + j < p.N) ? p.bias[n + j] : 0.f;
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[a][b][j] += bj;
      }
    }
  }
  if constexpr (NORM_EPI == 2) {
    // ---- RMSNorm + residual-add ADJOINT of the tile's rows (what fv_add_norm_bwd does with this GEMM's bf16 output as
    //      its d y; lane mapping and operation order of add_norm_bwd3_kernel<16>)
    static_assert(BN == 192 && WM * WN == 4 && BM % 32 == 0, "whole 192-wide rows, four waves");
    constexpr int RSB = BN * 2 + 16, LPR = 16, RPW = 4, RU = 2, RW = BM / 4;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const f32x4 v = acc[a][b];
        uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<uint2*>(smem + (wm * WMR + b * 16 + (lane & 15)) * RSB + (wn * WNC + a * 16 + (lane >> 4) * 4) * 2) = pk;
      }
    __syncthreads();
    FV_STAMP(ne, 3);                                   // product tile in LDS
    const int lr = lane % LPR, gr = lane / LPR;
    const float inv_n = 1.f / (float)BN;
    float w[3][4], aw[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(ne->w + (k * LPR + lr) * 4);
      w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) aw[k][e] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < RW / (RPW * RU); ++it) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int rl = wv * RW + it * (RPW * RU) + u * RPW + gr;
        const int row = m0 + rl;
        const bool live = row < mlim;
        const int rowc = live ? row : p.M - 1;
        const size_t base = (size_t)rowc * BN;
        const float rstd = ne->rstd[rowc];
        const float sc = ne->row_scale ? ne->row_scale[rowc / ne->rows_per_scale] : 1.f;
        const float lv = live ? 1.f : 0.f;
        float xh[3][4], dxh[3][4];
        float c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = (k * LPR + lr) * 4;
          const uint2 xb = *reinterpret_cast<const uint2*>(smem + rl * RSB + c * 2);
          const float dyv[4] = {__uint_as_float(xb.x << 16), __uint_as_float(xb.x & 0xffff0000u),
                                __uint_as_float(xb.y << 16), __uint_as_float(xb.y & 0xffff0000u)};
          float4 rr4;
          if constexpr (NE_PRE) rr4 = ne_r[it][u][k];
          else rr4 = *reinterpret_cast<const float4*>(ne->residual + base + c);
          const float rr[4] = {rr4.x, rr4.y, rr4.z, rr4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dyl = dyv[e] * lv;
            xh[k][e] = (rr[e] - 0.f) * rstd;
            dxh[k][e] = dyl * w[k][e];
            aw[k][e] = fmaf(dyl, xh[k][e], aw[k][e]);
            c2 = fmaf(dxh[k][e], xh[k][e], c2);
          }
        }
#define FV_DPP_ADD(ctrl) c2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c2), ctrl, 0xf, 0xf, true))
        FV_DPP_ADD(0xB1);
        FV_DPP_ADD(0x4E);
        FV_DPP_ADD(0x141);
        FV_DPP_ADD(0x140);
#undef FV_DPP_ADD
        c2 = c2 * inv_n;
        const float c1 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = (k * LPR + lr) * 4;
          float4 g4;
          if constexpr (NE_PRE) g4 = ne_g[it][u][k];
          else g4 = ne->dres_out ? *reinterpret_cast<const float4*>(ne->dres_out + base + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
          float dr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) dr[e] = rstd * (dxh[k][e] - c1 - xh[k][e] * c2) + gg[e];
          if (live) {
            *reinterpret_cast<float4*>(ne->res_out + base + c) = make_float4(dr[0], dr[1], dr[2], dr[3]);
            if (ne->row_scale)
#pragma unroll
              for (int e = 0; e < 4; ++e) dr[e] *= sc;
            uint2 pk = {pack_bf16x2(dr[0], dr[1]), pack_bf16x2(dr[2], dr[3])};
            *reinterpret_cast<uint2*>(ne->y + base + c) = pk;
            *reinterpret_cast<uint2*>(smem + rl * RSB + c * 2) = pk;      // d x stays in the tile for the second phase
          } else {
            *reinterpret_cast<uint2*>(smem + rl * RSB + c * 2) = make_uint2(0u, 0u);
          }
        }
      }
    }
    // d norm weight: the 4 row groups of a wave (permlane swaps), then the 4 waves through LDS, fixed order
    __syncthreads();
    FV_STAMP(ne, 4);                                   // norm adjoint of the rows done (stores issued)
    float* s_acc = reinterpret_cast<float*>(smem + BM * RSB);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = aw[k][e];
        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
        auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
        if (gr == 0) s_acc[wv * BN + (k * LPR + lr) * 4 + e] = v;
      }
    __syncthreads();
    float* dst = ne->pw + (size_t)block_id * BN;
    for (int c = tid; c < BN; c += NT) dst[c] = (s_acc[c] + s_acc[BN + c]) + (s_acc[2 * BN + c] + s_acc[3 * BN + c]);
    FV_STAMP(ne, 5);                                   // partial d weight written
    if (ne->W2) {
      // ---- second phase: the previous block's out_proj data gradient d g = d x @ W_out from the tile in LDS
      __syncthreads();
      constexpr int O_B = (BM * RSB + 255) / 256 * 256, O_S = O_B + 2 * 128 * BK * 2;
      tile_times_w2<BM, RSB, KS>(smem, smem + O_B, smem + O_S, ne->W2, ne->ldw2, ne->N2, ne->C2, m0, mlim, tid);
    }
    FV_STAMP(ne, 6);                                   // second phase done (its stores issued)
    return;
  }
  if constexpr (NORM_EPI == 1) {
    // ---- residual add + RMSNorm of the tile's rows (what fv_add_norm_fwd does to the bf16 GEMM output, same lane
    //      mapping and operation order as add_norm_fwd3_kernel<16>): the product is rounded to bf16 into an LDS tile,
    //      then a wave step takes 4 rows x 16 lanes x (3 x 4) channels.
    static_assert(BN == 192 && WM * WN == 4 && BM % 32 == 0, "whole 192-wide rows, four waves");
    constexpr int RSB = BN * 2 + 16, LPR = 16, RPW = 4, RU = 2, RW = BM / 4;   // RW rows per wave
    __syncthreads();                                  // all waves are done reading the operand tiles
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const f32x4 v = acc[a][b];
        uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<uint2*>(smem + (wm * WMR + b * 16 + (lane & 15)) * RSB + (wn * WNC + a * 16 + (lane >> 4) * 4) * 2) = pk;
      }
    __syncthreads();
    FV_STAMP(ne, 3);
    const int lr = lane % LPR, gr = lane / LPR;
    const float inv_n = 1.f / (float)BN;
    float w[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(ne->w + (k * LPR + lr) * 4);
      w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
    }
#pragma unroll
    for (int it = 0; it < RW / (RPW * RU); ++it) {
      float v[RU][3][4], r[RU][3][4], sc_u[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int rl = wv * RW + it * (RPW * RU) + u * RPW + gr;       // row of the tile
        const int row = m0 + rl, rowc = row < p.M ? row : p.M - 1;
        sc_u[u] = ne->row_scale ? ne->row_scale[rowc / ne->rows_per_scale] : 1.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = (k * LPR + lr) * 4;
          const uint2 xb = *reinterpret_cast<const uint2*>(smem + rl * RSB + c * 2);
          v[u][k][0] = __uint_as_float(xb.x << 16); v[u][k][1] = __uint_as_float(xb.x & 0xffff0000u);
          v[u][k][2] = __uint_as_float(xb.y << 16); v[u][k][3] = __uint_as_float(xb.y & 0xffff0000u);
          float4 t;
          if constexpr (NE_PRE) t = ne_r[it][u][k];
          else t = *reinterpret_cast<const float4*>(ne->residual + (size_t)rowc * BN + c);
          r[u][k][0] = t.x; r[u][k][1] = t.y; r[u][k][2] = t.z; r[u][k][3] = t.w;
        }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int row = m0 + wv * RW + it * (RPW * RU) + u * RPW + gr;
        const bool live = row < mlim;
        const size_t base = (size_t)(live ? row : p.M - 1) * BN;
        const float sc = sc_u[u];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = (k * LPR + lr) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][k][e] = fmaf(v[u][k][e], sc, r[u][k][e]);
          if (live) *reinterpret_cast<float4*>(ne->res_out + base + c) = make_float4(v[u][k][0], v[u][k][1], v[u][k][2], v[u][k][3]);
        }
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) q = fmaf(v[u][k][e], v[u][k][e], q);
        // 16-lane row sum: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
#define FV_DPP_ADD(ctrl) q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), ctrl, 0xf, 0xf, true))
        FV_DPP_ADD(0xB1);
        FV_DPP_ADD(0x4E);
        FV_DPP_ADD(0x141);
        FV_DPP_ADD(0x140);
#undef FV_DPP_ADD
        const float rstd = rsqrtf(q * inv_n + ne->eps);
        if (lr == 0 && live) ne->rstd[row] = rstd;
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
          if (live) *reinterpret_cast<uint2*>(ne->y + base + c) = pk;
          else pk = make_uint2(0u, 0u);
          // the normalised row replaces the product in the tile (same lane, same bytes) for the second phase
          *reinterpret_cast<uint2*>(smem + (wv * RW + it * (RPW * RU) + u * RPW + gr) * RSB + c * 2) = pk;
        }
      }
    }
    FV_STAMP(ne, 4);
    if (ne->W2) {
      // ---- second phase: this block's in_proj, xz = y @ W_in^T, from the normalised tile in LDS
      __syncthreads();
      constexpr int O_B = (BM * RSB + 255) / 256 * 256, O_S = O_B + 2 * 128 * BK * 2;
      tile_times_w2<BM, RSB, KC>(smem, smem + O_B, smem + O_S, ne->W2, ne->ldw2, ne->N2, ne->C2, m0, mlim, tid);
    }
    FV_STAMP(ne, 6);
    return;
  }
  if (!p.c_fp32) {
    // bf16 C: the wave's WMR x WNC tile goes through LDS (32-row slabs, rows padded by 16 B) so that
    // every global store is 16 B per lane and a wave instruction writes whole 128-byte row segments
    constexpr int RS = WNC * 2 + 16, CH = WNC / 8;
    __syncthreads();                                  // all waves are done reading the operand tiles
    char* my = smem + wv * (32 * RS);
    bf16_t* Cb = (bf16_t*)p.C + zoff;
#pragma unroll
    for (int h = 0; h < (MB + 1) / 2; ++h) {          // (an odd MB leaves a 16-row half slab at the end)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int b = 2 * h + bb;
        if (b >= MB) continue;
#pragma unroll
        for (int a = 0; a < NB; ++a) {
          const f32x4 v = acc[a][b];
          uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<uint2*>(my + (bb * 16 + (lane & 15)) * RS + (a * 16 + (lane >> 4) * 4) * 2) = pk;
        }
      }
      // the slab is private to the wave and LDS operations of one wave execute in order: no block barrier
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = i * 64 + lane, r = idx / CH, ch = idx - r * CH;
        const int m = m0 + wm * WMR + h * 32 + r, n = n0 + wn * WNC + ch * 8;
        if (m < p.M && n < p.N && h * 32 + r < WMR) {
          const u32x4 q = *reinterpret_cast<const u32x4*>(my + r * RS + ch * 16);
          bf16_t* dst = Cb + (long)m * p.ldc + n;
          if (n + 8 <= p.N && (((uintptr_t)dst) & 15) == 0) {
            *reinterpret_cast<u32x4*>(dst) = q;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)      // constant lane indices: no scratch copy of q
              if (n + j < p.N) reinterpret_cast<uint16_t*>(dst)[j] = (uint16_t)((j & 1) ? (q[j >> 1] >> 16) : (q[j >> 1] & 0xffffu));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    const int m = m0 + wm * WMR + b * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      const int n = n0 + wn * WNC + a * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[a][b];
      if (p.rb_period > 0) {      // what a bf16 F.linear followed by an fp32 add of a per-token table returns (patch embed)
        const float* t = p.bias + (long)(m % p.rb_period) * p.N + n;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < p.N) v[j] = bf16_bits_to_f32(f32_to_bf16_bits(v[j])) + t[j];
      }
      float* dst = (float*)p.C + zoff + (long)m * p.ldc + n;
      if (n + 3 < p.N) *reinterpret_cast<f32x4*>(dst) = v;
      else
        for (int j = 0; j < 4 && n + j < p.N; ++j) dst[j] = v[j];
    }
  }
}

template <int AMODE, int BMODE, int WM, int WN, bool GLDS, int NB = 4, int MB = 4>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bf16_kernel(GemmParams p) {
  gemm_bf16_body<AMODE, BMODE, WM, WN, GLDS, NB, MB>(p, blockIdx.x, blockIdx.z);
}

template <int BMROWS>
__global__ __launch_bounds__(256, 2) void gemm_addnorm_kernel(GemmParams p, NormEpi ne) {
  gemm_bf16_body<KC, KC, 2, 2, true, 6, BMROWS / 32, true, 1>(p, blockIdx.x, 0, &ne);
}

template <int BMROWS>
__global__ __launch_bounds__(256, 2) void gemm_dgrad_addnorm_bwd_kernel(GemmParams p, NormEpi ne) {
  gemm_bf16_body<KC, KS, 2, 2, true, 6, BMROWS / 32, true, 2>(p, blockIdx.x, 0, &ne);
}

// ---- phased 256 x 256 form of the forward GEMM (K-contiguous A and B, bf16 out) -----------------------------------
// The kernels above stage a K tile per __syncthreads() -- a vmcnt(0) with the next tile's LDS-DMA in flight -- and stop
// near 900 TFLOP/s at the FastVim-B widths whatever the tile shape.  Here the prefetch stays in
// flight across barriers: 8 waves as 2 (128-row halves) x 4 (64-column quarters), 128 KiB of LDS = 2 K tiles x {A lo, A hi,
// B lo, B hi} half tiles of 128 x 64, counted s_waitcnt vmcnt, raw s_barrier, LDS reads as inline asm (no memory operand:
// the compiler would order them behind every outstanding LDS-DMA load).  A K tile is four phases of 16 MFMAs per wave
// (one 64 x 32 quadrant of its 128 x 64 output over the tile's two 32-deep steps):
//     phase      reads (ds_read_b128)                 multiplies       LDS-DMA issued
//     P1(t)      A rows 0-63 [8], B cols 0-31 [4]     (m0, n0)         A lo, A hi of tile t+1
//     P2(t)      B cols 32-63 [4]                     (m0, n1)         --
//     P3(t)      A rows 64-127 [8]                    (m1, n1)         --
//     P4(t)      --   (B cols 0-31 kept)              (m1, n0)         B lo, B hi of tile t+2; then vmcnt: tile t+1 landed
// Each phase is  [reads, DMA issue, wait] s_barrier [lgkmcnt(0), MFMAs] s_barrier.  The row-half groups run one barrier
// apart (the second group takes one barrier up front, the first one at the end), so one group's MFMAs run under the
// other's reads.  What orders LDS-DMA against reads (MI355X guide, "pipelining across barriers"):
//   * a half tile is read no earlier than the phase AFTER the one whose wait retired it (both groups have then passed a
//     barrier behind every wave's wait);
//   * a half tile is refilled no earlier than TWO phases after its last read (the reader's lgkmcnt(0) sits behind the
//     barrier that ends its read section; the next barrier is behind that wait for both groups).
// B halves of tile t are last read in P2(t) and refilled in P4(t) (tile t+2, same parity); A halves are last read in P3(t)
// and refilled in P1(t+1).  Accumulation order per output element is the K order, as in the kernels above: bitwise the
// same C.
template <int NV_>
__device__ __forceinline__ void wait_vmc() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV_) : "memory"); }
template <int OFF>
__device__ __forceinline__ u32x4 ds_read_b128_imm(uint32_t a) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}

// AMODE / BMODE: KC = operand K-contiguous, KS = K-slow (row-major as stored, transposing LDS reads).  <KC, KC> forward,
// <KC, KS> data gradient, <KS, KS> weight gradient (CF32: fp32 partial of K slice `split`).  bid = tile of the problem.
// PERSIST (forward / data-gradient forms): the workgroup walks the tiles bid, bid + bid_stride, ... < bid_end.  The first
// K tiles of the NEXT output tile are requested before this one's C leaves (the stage buffers are free once every wave
// is past its last read; the C slabs of a persistent workgroup live in the 32 KiB above them, 16 rows at a time), so the
// next K loop starts on landed data instead of behind a workgroup launch, its address set-up and a cold first fetch.
template <int AMODE, int BMODE, bool CF32, bool PERSIST = false>
__device__ __forceinline__ void p256_body(const GemmParams& p, int bid, int split, char* smem, int bid_stride = 0,
                                          int bid_end = 0) {
  constexpr int HALF = 128 * BK * 2;            // bytes of a 128 x 64 half tile
  constexpr int PAR = 4 * HALF;                 // one K tile: A lo | A hi | B lo | B hi
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wv >> 2, wc = wv & 3;
  const int tiles_n = p.N / 256;
  int m0, n0;
  const int kbeg = split * p.k_per_split;
  const int nt = (min(p.K, kbeg + p.k_per_split) - kbeg) / BK;
  // LDS-DMA sources: thread -> physical 16-byte slot e = tid + 512 i of a half tile (rows 8 slots wide, chunk ^= row & 7)
  const int sr = tid >> 3, sc = (tid & 7) ^ (sr & 7);
  const bf16_t* srcA[4];
  const bf16_t* srcB[4];        // KC: as A;  KS: [half][piece] sources of the K-slow image ([k][128 cols], chunk-swizzled)
  auto set_tile = [&](int b_) {
    m0 = (b_ / tiles_n) * 256;
    n0 = (b_ % tiles_n) * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                 // row groups of 64: g = 2 half + i
      if constexpr (AMODE == KC) srcA[g] = p.A + (long)min(m0 + g * 64 + sr, p.M - 1) * p.lda + sc * 8 + kbeg;
      if constexpr (BMODE == KC) srcB[g] = p.B + (long)min(n0 + g * 64 + sr, p.N - 1) * p.ldb + sc * 8 + kbeg;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (AMODE == KS) {
        GldsPlan<KS, 128, 512> ga;
        ga.init(p.A, p.lda, m0 + h * 128, p.M, tid);
        srcA[2 * h] = ga.src[0] + (long)kbeg * p.lda;
        srcA[2 * h + 1] = ga.src[1] + (long)kbeg * p.lda;
      }
      if constexpr (BMODE == KS) {
        GldsPlan<KS, 128, 512> gb;
        gb.init(p.B, p.ldb, n0 + h * 128, p.N, tid);
        srcB[2 * h] = gb.src[0] + (long)kbeg * p.ldb;
        srcB[2 * h + 1] = gb.src[1] + (long)kbeg * p.ldb;
      }
    }
  };
  set_tile(bid);
  typedef __attribute__((address_space(1))) const void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  auto issueA = [&](int t) {
    char* base = smem + (t & 1) * PAR;
#ifdef FASTVIM_TUNING_HOOKS
    if (p.rb_period & 16) return;
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_global_load_lds((gptr)(srcA[g] + (long)t * BK * (AMODE == KC ? 1 : p.lda)),
                                       (lptr)(base + (g >> 1) * HALF + ((g & 1) * 512 + wv * 64) * 16), 16, 0, 0);
  };
  auto issueB = [&](int t) {
    char* base = smem + (t & 1) * PAR + 2 * HALF;
#ifdef FASTVIM_TUNING_HOOKS
    if (p.rb_period & 16) return;
#endif
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_global_load_lds((gptr)(srcB[g] + (long)t * BK * (BMODE == KC ? 1 : p.ldb)),
                                       (lptr)(base + (g >> 1) * HALF + ((g & 1) * 512 + wv * 64) * 16), 16, 0, 0);
  };
  // fragment addresses: row (lane & 15) of a 16-row block, 16-byte chunk ks * 4 + (lane >> 4), swizzled by row & 7
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
  const uint32_t off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);     // ks = 0; ks = 1: ^ 64
  const uint32_t aA = lds0 + wr * HALF + off0;                                         // + parity * PAR + blk * 2048
  const uint32_t aB = lds0 + 2 * HALF + (wc >> 1) * HALF + (wc & 1) * (64 * 128) + off0;
  // K-slow operands: one address per 16-wide block (transposing reads, see KsFrags): the 8 row blocks of the wave's A half,
  // the 4 column blocks of its 64 columns
  uint32_t aAk[8];
  if constexpr (AMODE == KS) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, kl = g * 8 + q;
#pragma unroll
    for (int j = 0; j < 8; ++j) aAk[j] = lds0 + wr * HALF + kl * 256 + ((j ^ ks_swz<128>(kl)) << 5) + pp * 8;
  }
  uint32_t aBk[4];
  if constexpr (BMODE == KS) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, kl = g * 8 + q;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      aBk[j] = lds0 + 2 * HALF + (wc >> 1) * HALF + kl * 256 + ((((wc & 1) * 4 + j) ^ ks_swz<128>(kl)) << 5) + pp * 8;
  }

  f32x4 acc[4][8];       // [n tile][m tile]
  u32x4 fA[2][4], fB0[2][2], fB1[2][2];          // [ks][block]
  unsigned long long kB0[2][2][2], kB1[2][2][2];  // K-slow B: [ks][block][lo | hi], joined behind the wait
  unsigned long long kA[2][4][2];                 // K-slow A likewise
#ifdef FASTVIM_TUNING_HOOKS
  const int dbgf = p.rb_period;                  // phase probe: 1 no stores, 2 no vmcnt waits, 4 no MFMAs, 8 no LDS reads
  if (dbgf & 8) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fA[ks][i] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 2; ++i) fB0[ks][i] = fB1[ks][i] = (u32x4){0u, 0u, 0u, 0u};
    }
  }
#else
  constexpr int dbgf = 0;
#endif

  bool first_tile = true;
  for (;;) {        // output tiles of a persistent workgroup (one pass otherwise)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // prologue: tiles 0 and 1 in flight, tile 0 landed
  if (!PERSIST || first_tile) {
    issueA(0); issueB(0);
    if (nt > 1) { issueA(1); wait_vmc<4>(); } else wait_vmc<0>();
  } else {
    wait_vmc<0>();      // requested before the previous tile's C stores; those stores have completed as well
  }
  __builtin_amdgcn_s_barrier();
  if (nt > 1) issueB(1);
  if (wr == 1) __builtin_amdgcn_s_barrier();      // the second row-half group runs one barrier behind

#define FV_RD_A(PARITY, MH)                                                                                  \
  if (!(dbgf & 8)) static_for<4>([&](auto mt) {                                                              \
    if constexpr (AMODE == KC) {                                                                             \
      fA[0][mt] = ds_read_b128_imm<(MH * 4 + mt) * 2048>(aA + (PARITY) * PAR);                               \
      fA[1][mt] = ds_read_b128_imm<(MH * 4 + mt) * 2048>((aA + (PARITY) * PAR) ^ 64u);                       \
    } else {                                                                                                 \
      const uint32_t a_ = aAk[MH * 4 + mt] + (PARITY) * PAR;                                                 \
      kA[0][mt][0] = ds_read_tr16_b64<0>(a_);                                                                \
      kA[0][mt][1] = ds_read_tr16_b64<4 * 256>(a_);                                                          \
      kA[1][mt][0] = ds_read_tr16_b64<32 * 256>(a_);                                                         \
      kA[1][mt][1] = ds_read_tr16_b64<32 * 256 + 4 * 256>(a_);                                               \
    }                                                                                                        \
  })
#define FV_RD_B(DST, KDST, PARITY, NH)                                                                       \
  if (!(dbgf & 8)) static_for<2>([&](auto nn) {                                                              \
    if constexpr (BMODE == KC) {                                                                             \
      DST[0][nn] = ds_read_b128_imm<(NH * 2 + nn) * 2048>(aB + (PARITY) * PAR);                              \
      DST[1][nn] = ds_read_b128_imm<(NH * 2 + nn) * 2048>((aB + (PARITY) * PAR) ^ 64u);                      \
    } else {                                                                                                 \
      const uint32_t a_ = aBk[NH * 2 + nn] + (PARITY) * PAR;                                                 \
      KDST[0][nn][0] = ds_read_tr16_b64<0>(a_);                                                              \
      KDST[0][nn][1] = ds_read_tr16_b64<4 * 256>(a_);                                                        \
      KDST[1][nn][0] = ds_read_tr16_b64<32 * 256>(a_);                                                       \
      KDST[1][nn][1] = ds_read_tr16_b64<32 * 256 + 4 * 256>(a_);                                             \
    }                                                                                                        \
  })
#define FV_MM(BF, KBF, MH, NH)                                                                               \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                     \
        if constexpr (AMODE == KC) { asm volatile("" : "+v"(fA[ks][mt])); }                                  \
        else {                                                                                               \
          asm volatile("" : "+v"(kA[ks][mt][0]));                                                            \
          asm volatile("" : "+v"(kA[ks][mt][1]));                                                            \
          typedef unsigned long long u64x2a_ __attribute__((ext_vector_type(2)));                            \
          const u64x2a_ j_ = {kA[ks][mt][0], kA[ks][mt][1]};                                                 \
          fA[ks][mt] = __builtin_bit_cast(u32x4, j_);                                                        \
        }                                                                                                    \
      }                                                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int nn = 0; nn < 2; ++nn) {                                                     \
        if constexpr (BMODE == KC) { asm volatile("" : "+v"(BF[ks][nn])); }                                  \
        else {                                                                                               \
          asm volatile("" : "+v"(KBF[ks][nn][0]));                                                           \
          asm volatile("" : "+v"(KBF[ks][nn][1]));                                                           \
          typedef unsigned long long u64x2_ __attribute__((ext_vector_type(2)));                             \
          const u64x2_ j_ = {KBF[ks][nn][0], KBF[ks][nn][1]};                                                \
          BF[ks][nn] = __builtin_bit_cast(u32x4, j_);                                                        \
        }                                                                                                    \
      }                                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if (!(dbgf & 4))                                                                                         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int nn = 0; nn < 2; ++nn)                                                       \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                     \
          acc[NH * 2 + nn][MH * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                           \
              __builtin_bit_cast(bf16x8, BF[ks][nn]), __builtin_bit_cast(bf16x8, fA[ks][mt]),                \
              acc[NH * 2 + nn][MH * 4 + mt], 0, 0, 0);                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)

  for (int t = 0; t < nt; ++t) {
    const uint32_t par = t & 1;
    // P1
    FV_RD_A(par, 0);
    FV_RD_B(fB0, kB0, par, 0);
    if (t >= 1 && t + 1 < nt) issueA(t + 1);       // (tile 1's A went out in the prologue)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    FV_MM(fB0, kB0, 0, 0);
    __builtin_amdgcn_s_barrier();
    // P2
    FV_RD_B(fB1, kB1, par, 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    FV_MM(fB1, kB1, 0, 1);
    __builtin_amdgcn_s_barrier();
    // P3
    FV_RD_A(par, 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    FV_MM(fB1, kB1, 1, 1);
    __builtin_amdgcn_s_barrier();
    // P4: B of tile t + 2 into this tile's B halves (last read in P2); everything older -- tile t + 1 -- has landed
#ifdef FASTVIM_TUNING_HOOKS
    if (p.rb_period & 2) { if (t + 2 < nt) issueB(t + 2); } else
#endif
    if (t + 2 < nt) { issueB(t + 2); wait_vmc<4>(); } else wait_vmc<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    FV_MM(fB0, kB0, 1, 0);
    __builtin_amdgcn_s_barrier();
  }
#undef FV_MM
#undef FV_RD_B
#undef FV_RD_A
  if (wr == 0) __builtin_amdgcn_s_barrier();
  // epilogue: bf16 through a wave-private LDS slab (64 rows x 64 columns at a time), 16-byte stores of whole 128-byte
  // row segments.  Every wave is past its last LDS read and every LDS-DMA load has landed (vmcnt(0) in the last P4).
  if constexpr (CF32) {
    // fp32 partial of this K slice: 16 bytes per lane (64-byte row segments); with thousands of K tiles per output tile
    // the epilogue does not matter
    float* Cf = (float*)p.C + (long)split * p.c_split_stride;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int m = m0 + wr * 128 + b * 16 + (lane & 15), n = n0 + wc * 64 + a * 16 + (lane >> 4) * 4;
        if (m < p.M) {
          f32x4* dst = reinterpret_cast<f32x4*>(Cf + (long)m * p.ldc + n);
          // c_fp32 == 2: the one K slice of the problem is added to what C holds (a weight gradient accumulated in place:
          // every element belongs to one workgroup, so the sum is as deterministic as the partial + reduction it replaces)
          *dst = p.c_fp32 == 2 ? *dst + acc[a][b] : acc[a][b];
        }
      }
    return;
  }
  __syncthreads();
  constexpr int RS = 64 * 2 + 16;
  bf16_t* C = (bf16_t*)p.C;
  if constexpr (PERSIST) {
    const int m0e = m0, n0e = n0;
    const bool more = bid + bid_stride < bid_end;
    if (more) {          // the next tile's first K tiles go out under this tile's stores
      bid += bid_stride;
      set_tile(bid);
      issueA(0); issueB(0);
      if (nt > 1) issueA(1);
    }
    char* my = smem + 2 * PAR + wv * (16 * RS);        // above the stage buffers: 16 rows x 64 columns per wave and pass
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 v = acc[a][bb];
        uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<uint2*>(my + (lane & 15) * RS + (a * 16 + (lane >> 4) * 4) * 2) = pk;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = i * 64 + lane, r = idx >> 3, ch = idx & 7;
        const int m = m0e + wr * 128 + bb * 16 + r, n = n0e + wc * 64 + ch * 8;
#ifdef FASTVIM_TUNING_HOOKS
        if (p.rb_period & 1) continue;
#endif
        if (m < p.M) *reinterpret_cast<u32x4*>(C + (long)m * p.ldc + n) = *reinterpret_cast<const u32x4*>(my + r * RS + ch * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!more) break;
    first_tile = false;
    continue;
  }
  char* my = smem + wv * (64 * RS);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 v = acc[a][h * 4 + b];
        uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<uint2*>(my + (b * 16 + (lane & 15)) * RS + (a * 16 + (lane >> 4) * 4) * 2) = pk;
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane, r = idx >> 3, ch = idx & 7;
      const int m = m0 + wr * 128 + h * 64 + r, n = n0 + wc * 64 + ch * 8;
#ifdef FASTVIM_TUNING_HOOKS
      if (p.rb_period & 1) continue;
#endif
      if (m < p.M) *reinterpret_cast<u32x4*>(C + (long)m * p.ldc + n) = *reinterpret_cast<const u32x4*>(my + r * RS + ch * 16);
    }
    __builtin_amdgcn_wave_barrier();
  }
  break;
  }      // output tiles
}

template <int BMODE>
__global__ __launch_bounds__(512, 2) void gemm_nt256p_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = ((p.M + 255) / 256) * (p.N / 256);
  int bid = blockIdx.x;
  {      // XCD-aware order: the tiles of one A row panel are consecutive on one XCD
    const int q8 = nblk / 8, r8 = nblk % 8, xcd = bid % 8, j = bid / 8;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
  }
  p256_body<KC, BMODE, false>(p, bid, 0, smem);
}

// persistent form: gridDim.x (a multiple of 8: one workgroup per CU) workgroups; the one on XCD x, slot j walks the tiles
// j, j + G/8, ... of that XCD's contiguous eighth of the tile order above
template <int BMODE>
__global__ __launch_bounds__(512, 2) void gemm_nt256pp_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nblk = ((p.M + 255) / 256) * (p.N / 256);
  const int w = blockIdx.x, per = gridDim.x / 8;
  const int q8 = nblk / 8, r8 = nblk % 8, xcd = w % 8, j = w / 8;
  const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, cnt = q8 + (xcd < r8 ? 1 : 0);
  if (j >= cnt) return;
  p256_body<KC, BMODE, false, true>(p, start + j, 0, smem, per, start + cnt);
}

// Several independent problems in ONE launch (the weight gradients of a whole backward pass, queued until its end):
// a single weight-gradient GEMM at FastVim-T is 168-336 workgroups -- a third to two thirds of what the chip holds at
// once -- and every launch pays that tail; the grouped launch is one long queue of workgroups.
constexpr int GROUP_MAX = 40;      // 40 x 88-byte problems + prefix table stay under the 4 KiB kernel-argument limit
struct GroupedParams {
  GemmParams p[GROUP_MAX];
  int blk_end[GROUP_MAX];     // exclusive prefix of (tiles x splits) workgroups per problem
  int count;
};

template <int AMODE, int BMODE, int WM, int WN, bool GLDS, int NB = 4, int MB = 4>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bf16_grouped_kernel(GroupedParams G, int xcd_order) {
  // Hardware block g runs on XCD g % 8.  The workgroups of one (problem, K slice) read the same token range of both
  // operands: in the logical order [problem][slice][tile] they are neighbours, so XCD x takes the x-th eighth of that
  // order (neighbours share an XCD and its L2, and start together) instead of every eighth workgroup -- with the
  // round-robin order the 12 tiles of an in_proj slice sat on 8 XCDs and each fetched its operands over the fabric
  int g = blockIdx.x;
  if (xcd_order & 255) {
    const int n = gridDim.x, q8 = n / 8, r8 = n % 8, xcd = g % 8, k = g / 8;
    g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  int j = 0;
  while (j + 1 < G.count && g >= G.blk_end[j]) ++j;
  const GemmParams& p = G.p[j];
  const int local = g - (j ? G.blk_end[j - 1] : 0);
  constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN;
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const int split = local / tiles;
  gemm_bf16_body<AMODE, BMODE, WM, WN, GLDS, NB, MB, false>(p, local - split * tiles, split, nullptr, xcd_order >> 8);
}

// the phased 256 x 256 form for the weight gradients of the large outputs (M and N multiples of 256: FastVim-B)
__global__ __launch_bounds__(512, 2) void gemm_p256_grouped_kernel(GroupedParams G, int xcd_order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int g = blockIdx.x;
  if (xcd_order & 255) {
    const int n = gridDim.x, q8 = n / 8, r8 = n % 8, xcd = g % 8, k = g / 8;
    g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  int j = 0;
  while (j + 1 < G.count && g >= G.blk_end[j]) ++j;
  const GemmParams& p = G.p[j];
  const int local = g - (j ? G.blk_end[j - 1] : 0);
  const int tiles = (p.M / 256) * (p.N / 256);
  const int split = local / tiles;
  p256_body<KS, KS, true>(p, local - split * tiles, split, smem);
}

template <int AMODE, int BMODE, int WM, int WN, bool GLDS, int NB = 4, int MB = 4>
int launch_k(const GemmParams& p, int splits, hipStream_t st) {
  constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN;
  const int tiles = fv_cdiv(p.M, BM) * fv_cdiv(p.N, BN);
  const size_t smem = (size_t)2 * (BM + BN) * BK * 2;
  static FvOncePerDevice attr_set;   
  if (smem > 64 * 1024 && attr_set.first()) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AMODE, BMODE, WM, WN, GLDS, NB, MB>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)0;         
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<AMODE, BMODE, WM, WN, GLDS, NB, MB>), dim3(tiles, 1, splits), dim3(64 * WM * WN), smem, st, p);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <int AMODE, int BMODE, int WM, int WN>
int launch(const GemmParams& p, int splits, hipStream_t st) {
  // LDS-DMA staging needs whole 64-deep K tiles in every split and >= 8 columns in K-slow operands
  static const bool allow = (fv_tune("FASTVIM_GEMM_GLDS", 1) != 0);   // tuning hook
  // (measured: forward and data-gradient GEMMs -8..-20 %; the K-slow x K-slow weight-gradient form lost 10 % while its
  //  transposing reads were compiler-visible -- every K step waited for all LDS-DMA loads -- and gains 15 % with the
  //  opaque immediate-offset reads of KsFrags)
  static const bool ks_glds = (fv_tune("FASTVIM_WGRAD_GLDS", 1) != 0);   // tuning hook
  const bool whole = (AMODE == KC || (ks_glds && p.M >= 8)) && p.K % BK == 0 && p.k_per_split % BK == 0 &&
                     (BMODE == KC || p.N >= 8);
  if (allow && whole) return launch_k<AMODE, BMODE, WM, WN, true>(p, splits, st);
  return launch_k<AMODE, BMODE, WM, WN, false>(p, splits, st);
}


// ---- streaming form of the forward / data-gradient GEMMs ------------------------------------------------------
// At the FastVim-T widths (K = 192 .. 768) a 128-row tile has 3 .. 12 K steps, and a workgroup of the kernel above
// lives mostly in start-up and drain: with the phases switched off one at a time (tools/gemm_phase_probe.py,
// HBM-cold), in_proj forward takes 24.5 us of which 10.8 us remain with no MFMA, no stores and a single staged K
// step -- cold kernel arguments, cold instructions, the first load's full latency, an epilogue with nothing in
// flight.  Here ONE persistent 8-wave workgroup per CU walks its tiles as a flat sequence of (tile, K step) stages
// through an S-deep LDS ring: S-1 stages are in flight (s_waitcnt vmcnt(n) on the oldest only -- loads return in
// order), the first stages of the next tile are issued before the last ones of this tile are multiplied, and the
// epilogue stores (wave-private LDS slab -> 16-byte rows) run under them.  Two waves per SIMD matter: with four
// waves (one per SIMD) the ~130 instructions of a stage run at single-wave issue latency and the same design was
// 20-80 % SLOWER than the kernel above.  rows_per_tile <= BM trims the tile height so that every workgroup gets
// the same number of tiles; the trimmed rows are computed but not stored.  Measured gain is small (HBM-cold:
// in_proj forward 24.3 -> 23.0 us, in_proj data gradient 20.4 -> 18.3 us; inside the step the data gradients gain
// 14 %, the forward GEMMs lose 6 % and keep the per-tile kernel): ring depth 2 vs 3 makes no difference, the fixed
// ~5-9 us of a launch at these sizes does.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BMODE, int WM, int WN, int NB, int MB, int S>
__global__ __launch_bounds__(64 * WM * WN) void gemm_stream_kernel(GemmParams p, int rpt, int tiles_m, int tiles_n) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN, WNC = 16 * NB, WMR = 16 * MB;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int RS = WNC * 2 + 16, CH = WNC / 8, SLAB = 16 * RS;
  constexpr int NL = GldsPlan<KC, BM, NT>::NV + GldsPlan<BMODE, BN, NT>::NV;      // loads per thread per stage
  static_assert((S - 2) * NL < 64, "vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / WN, wn = wv % WN;
  char* my = smem + S * STAGE + wv * SLAB;
  const uint32_t my_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)my;
  const int nblk = tiles_m * tiles_n, G = gridDim.x;
  const int count = (nblk - (int)blockIdx.x + G - 1) / G;
  const int nt = p.K / BK, total = count * nt;
  const bool remap = (G & 7) == 0;
  const int q8 = nblk / 8, r8 = nblk % 8;
  // tile i of this workgroup.  Workgroup w runs on XCD w % 8; so does virtual block w + i * G: the XCD-aware order of
  // the kernel above (the N tiles sharing an A row panel are neighbours on one XCD) carries over
  auto origin = [&](int i, int& m0, int& n0) {
    int v = blockIdx.x + i * G;
    if (remap) {
      const int xcd = v & 7, j = v >> 3;
      v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    const int tm = v / tiles_n;
    m0 = tm * rpt;
    n0 = (v - tm * tiles_n) * BN;
  };

  GldsPlan<KC, BM, NT> ga;
  GldsPlan<BMODE, BN, NT> gb;
  int iq = 0, ikt = 0, itile = 0, islot = 0;
  auto issue_next = [&]() {
    if (iq < total) {
      if (ikt == 0) {
        int m0, n0;
        origin(itile, m0, n0);
        ga.init(p.A, p.lda, m0, p.M, tid);
        gb.init(p.B, p.ldb, n0, p.N, tid);
      }
      char* st = smem + islot * STAGE;
      ga.issue(st, ikt * BK, tid);
      gb.issue(st + A_BYTES, ikt * BK, tid);
      if (++ikt == nt) { ikt = 0; ++itile; }
      if (++islot == S) islot = 0;
    }
    ++iq;
  };

  f32x4 acc[NB][MB];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  KsFrags<BN, BMODE == KS ? NB : 1> kb;
  if constexpr (BMODE == KS) kb.init(smem + A_BYTES, wn * NB, lane);
#pragma unroll
  for (int j = 0; j < S - 1; ++j) issue_next();
  int kt = 0, tile = 0, slot = 0;
  for (int s = 0; s < total; ++s) {
    // stage s has landed once at most the stages issued after it are outstanding
    const int later = min(S - 2, total - 1 - s);
    if (S >= 4 && later >= 2) wait_vm<(S >= 4 ? 2 : 0) * NL>();
    else if (S >= 3 && later >= 1) wait_vm<(S >= 3 ? 1 : 0) * NL>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();       // everyone's pieces of stage s are in LDS; everyone is done with stage s - 1
    asm volatile("" ::: "memory");
    issue_next();                       // stage s + S - 1 -> the slot stage s - 1 was read from
    const char* sA = smem + slot * STAGE;
    const char* sB = sA + A_BYTES;
    if constexpr (BMODE == KS) {
      kb.read(slot * STAGE);
      kb.wait();
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 fa[MB], fb[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) fa[i] = frag<KC, BM>(sA, wm * MB + i, ks, lane);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if constexpr (BMODE == KS) fb[i] = kb.get(ks, i);
        else fb[i] = frag<BMODE, BN>(sB, wn * NB + i, ks, lane);
      }
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[a], fa[b], acc[a][b], 0, 0, 0);
    }
    if (++slot == S) slot = 0;
    if (++kt < nt) continue;
    // ---- tile finished: bias, bf16, out through the wave's slab (16 rows at a time) ----
    kt = 0;
    int m0, n0;
    origin(tile++, m0, n0);
    const int mlim = min(p.M, m0 + rpt);
    if (p.bias) {
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const int n = n0 + wn * WNC + a * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bj = (n + j < p.N) ? p.bias[n + j] : 0.f;
#pragma unroll
          for (int b = 0; b < MB; ++b) acc[a][b][j] += bj;
        }
      }
    }
    bf16_t* Cb = (bf16_t*)p.C;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const f32x4 v = acc[a][b];
        const unsigned long long pk = (unsigned long long)pack_bf16x2(v[0], v[1]) | ((unsigned long long)pack_bf16x2(v[2], v[3]) << 32);
        // written with an opaque ds_write: the compiler orders every LDS store it can see behind ALL outstanding
        // LDS-DMA loads (s_waitcnt vmcnt(0)), which would drain the ring at every tile end; the slab is disjoint
        // from the ring and private to this wave, and the LDS queue of a wave is in order
        asm volatile("ds_write_b64 %0, %1" ::"v"(my_lds + (uint32_t)((lane & 15) * RS + (a * 16 + (lane >> 4) * 4) * 2)), "v"(pk) : "memory");
        acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < (16 * CH + 63) / 64; ++i) {
        const int idx = i * 64 + lane, r = idx / CH, ch = idx - r * CH;
        const int m = m0 + wm * WMR + b * 16 + r, n = n0 + wn * WNC + ch * 8;
        if (idx < 16 * CH && m < mlim && n < p.N) {
          const u32x4 q = *reinterpret_cast<const u32x4*>(my + r * RS + ch * 16);
          bf16_t* dst = Cb + (long)m * p.ldc + n;
          if (n + 8 <= p.N && (((uintptr_t)dst) & 15) == 0) {
            *reinterpret_cast<u32x4*>(dst) = q;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)      // constant lane indices: no scratch copy of q
              if (n + j < p.N) reinterpret_cast<uint16_t*>(dst)[j] = (uint16_t)((j & 1) ? (q[j >> 1] >> 16) : (q[j >> 1] & 0xffffu));
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int BMODE, int WM, int WN, int NB, int MB, int S>
int launch_stream(const GemmParams& p, hipStream_t st) {
  constexpr int BM = 16 * MB * WM, BN = 16 * NB * WN;
  constexpr size_t smem = (size_t)S * (BM + BN) * BK * 2 + (size_t)WM * WN * 16 * (16 * NB * 2 + 16);
  const int cus = fv_cu_count();
  static FvOncePerDevice attr_set;
  if (attr_set.first())
    (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<BMODE, WM, WN, NB, MB, S>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  static const int per_cu_env = fv_tune("FASTVIM_GEMM_STREAM_WG", 0);   // tuning hook
  static const bool balance = (fv_tune("FASTVIM_GEMM_STREAM_BAL", 1) != 0);   // tuning hook
  const int fit = (int)((160 * 1024) / smem);
  const int per_cu = per_cu_env > 0 ? (per_cu_env < fit ? per_cu_env : fit) : fit;
  int G = cus * (per_cu < 1 ? 1 : per_cu);
  const int tiles_n = fv_cdiv(p.N, BN);
  int tiles_m = fv_cdiv(p.M, BM), rpt = BM;
  const int total0 = tiles_m * tiles_n;
  if (total0 <= G) {
    G = total0;
  } else if (balance) {
    const int rounds = fv_cdiv(total0, G), want_m = rounds * G / tiles_n;
    rpt = fv_cdiv(p.M, want_m);
    if (rpt > BM) rpt = BM;
    tiles_m = fv_cdiv(p.M, rpt);
  }
  hipLaunchKernelGGL((gemm_stream_kernel<BMODE, WM, WN, NB, MB, S>), dim3(G), dim3(64 * WM * WN), smem, st, p, rpt,
                     tiles_m, tiles_n);
  FV_LAUNCH_CHECK();
  return FV_OK;
}

template <int AMODE, int BMODE>
int launch_shape(const GemmParams& p, int splits, hipStream_t st) {
  // N a multiple of 192 (FastVim-T/S/B: d, 2*d_in): 128x192 tiles, the A panel is read N/192 times instead of
  // N/128.  Measured on MI355X: N = 384 -9 %, FastVim-S/B steps -3 %; N = 192 neutral (K-contiguous B) or +12 %
  // (K-slow B, 4-way swizzle), N = 768 with K = 192 +3 % -- those keep the 128-wide tiles.
  static const int tall = fv_tune("FASTVIM_GEMM_TALL", 0);   // tuning hook: 1 = 256x128, 2 = 256x192, 3 = 256x256 tiles
  const bool whole_k = p.K % BK == 0 && p.k_per_split % BK == 0;
  // streaming form (gemm_stream_kernel): K-contiguous activations, bf16 output, no split-K, the FastVim-T weight
  // sizes it was measured on (HBM-cold, M = 25088: in_proj forward 24.3 -> 23.0 us, in_proj data gradient
  // 20.4 -> 18.3; whole step 7.635 -> 7.59 ms)
  static const bool stream = (fv_tune("FASTVIM_GEMM_STREAM", 1) != 0);   // tuning hook
  if constexpr (AMODE == KC) {
    // (K-slow B only: inside the training step, where A was written by the previous kernel, the <KC, KC> forward
    //  GEMMs measured 21.1 -> 22.4 ms per 27 steps with it, the <KC, KS> data gradients 23.8 -> 20.1 ms)
    if (stream && BMODE == KS && !tall && whole_k && splits == 1 && !p.c_fp32 && p.M >= 8192 && p.K >= 128 &&
        p.N % 192 == 0 && (long)p.N * p.K <= 192 * 768)
      return launch_stream<BMODE, 2, 4, 3, 4, 3>(p, st);
  }
#ifdef FASTVIM_TUNING_HOOKS      // tile-shape experiments that lost their A/B: tuning builds only
  if (tall && p.M >= 256 && p.N >= 128 && (AMODE != KC || whole_k)) {
    constexpr bool G = AMODE == KC;
    if (tall == 1) return launch_k<AMODE, BMODE, 2, 2, G, 4, 8>(p, splits, st);
    if (tall == 2 && p.N % 192 == 0) return launch_k<AMODE, BMODE, 2, 2, G, 6, 8>(p, splits, st);
    if (tall == 3) return launch_k<AMODE, BMODE, 2, 2, G, 8, 8>(p, splits, st);
    if (tall == 4) return launch_k<AMODE, BMODE, 4, 2, G, 8, 4>(p, splits, st);     // 8 waves of 64x128: 256x256
    if (tall == 5) return launch_k<AMODE, BMODE, 2, 4, G, 4, 8>(p, splits, st);     // 8 waves of 128x64: 256x256
    if (tall == 6) return launch_k<AMODE, BMODE, 4, 2, G, 4, 4>(p, splits, st);     // 8 waves of 64x64: 256x128
    if (tall == 7 && p.N % 192 == 0) return launch_k<AMODE, BMODE, 4, 2, G, 6, 4>(p, splits, st);   // 8 waves of 64x96: 256x192
  }
#endif
  // phased 256 x 256 kernel (gemm_nt256p_kernel): one workgroup per CU, so it wants whole rounds of 256 tiles -- from
  // two rounds on.  Measured against the kernels below (same C, bit for bit): M = 131072 (2048 px, bs 8) N = 3072
  // K = 768 684 -> 608 us, N = 768 K = 1536 360 -> 331; M = 100352 (channel model) N = 1536 K = 384 188 -> 158;
  // M = 25088 N = 3072 132 -> 120-128 (4.6 rounds); M = 25088 N = 768 (1.15 rounds) 70.9 -> 72.9: stays below.
  // Data gradients (B as stored, transposing reads): M = 131072 N = 768 K = 3072 613 -> 547, N = 1536 K = 768 397 -> 341
  // fewest tiles it takes: two rounds (FastVim-B 224 px out_proj data gradient, 588 tiles: step 30.53 -> 30.36 ms)
  static const int p256_min = fv_tune("FASTVIM_GEMM_P256_MIN", 2 * 256);   // tuning hook
  static const int phased = fv_tune("FASTVIM_GEMM_P256", 3);   // tuning hook: bit 0 forward (<KC, KC>), bit 1 data gradient (<KC, KS>)
  // (short K loops lose: K = 128 / 192 forward +2-8 %, K = 384 data gradient even, K = 192 data gradient +8 %)
  if ((phased & (BMODE == KC ? 1 : 2)) && AMODE == KC && p.N % 256 == 0 && p.K % BK == 0 && p.K >= (BMODE == KC ? 384 : 512) && splits == 1 &&
      !p.c_fp32 && !p.bias && p.ldc % 8 == 0 && (long)fv_cdiv(p.M, 256) * (p.N / 256) >= p256_min) {
    static FvOncePerDevice attr;   
    if (attr.first()) {
      (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<BMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      (void)0;     
    }
    GemmParams q = p;
    q.rb_period = fv_tune("FASTVIM_GEMM_P256_DBG", 0);      // phase probe (tuning builds)
    // persistent workgroups, one per CU (round 3): the next tile's first K tiles are requested before a tile's C leaves.
    // Same C bit for bit; same box: FastVim-B 224 px 29.04 -> 28.75 ms per step, 2048 px 109.5 -> 107.9, FastChannelVim-S
    // 44.48 -> 44.30; stand-alone 2-5 % per GEMM (profiles/r03_p256_persistent.log)
    static const int persist = fv_tune("FASTVIM_GEMM_P256_PERSIST", 1);   // tuning hook: 0 = one workgroup per tile
    const int ntile = fv_cdiv(p.M, 256) * (p.N / 256);
    const int cus8 = fv_cu_count() / 8 * 8;
    static FvOncePerDevice attr2;
    if (attr2.first())
      (void)hipFuncSetAttribute((const void*)gemm_nt256pp_kernel<BMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (persist && cus8 >= 8 && ntile > cus8) {
      hipLaunchKernelGGL(gemm_nt256pp_kernel<BMODE>, dim3(cus8), dim3(512), 160 * 1024, st, q);
      FV_LAUNCH_CHECK();
      return FV_OK;
    }
    hipLaunchKernelGGL(gemm_nt256p_kernel<BMODE>, dim3(ntile), dim3(512), 128 * 1024, st, q);
    FV_LAUNCH_CHECK();
    return FV_OK;
  }
  // FastVim-B in_proj forward (N = 3072, K = 768): eight waves of 128x64 on a 256x256 tile, -10 % (161 -> 144 us);
  // measured slower at every FastVim-T/S shape and for the data-gradient forms, which keep the 4-wave tiles
  static const bool big = (fv_tune("FASTVIM_GEMM_BIG", 1) != 0);   // tuning hook
  if (big && !tall && AMODE == KC && BMODE == KC && p.N % 256 == 0 && p.N >= 2048 && p.K >= 512 && p.M >= 4096 && whole_k)
    return launch_k<AMODE, BMODE, 2, 4, true, 4, 8>(p, splits, st);
  // 160 x 128 tiles where they save rounds of workgroups: the per-tile launches run in rounds of 512 resident workgroups
  // (two per CU), so a launch costs about rounds x tile area.  FastVim-T in_proj forward (M = 25 088, N = 768, K = 192):
  // 1 176 tiles of 128 x 128 = 2.3 -> 3 rounds, 942 tiles of 160 x 128 = 1.84 -> 2 rounds at 1.25 x the area: 3.0 -> 2.5
  // (step 5.343 -> 5.305 ms, same box); FastVim-B out_proj forward (N = 768, K = 1536): 3.0 (either 128-row shape) -> 2.5;
  // S-width out_proj forward (N = 384): one round of 128 x 192 = 1.5 -> one round of 160 x 128 = 1.25.  The data gradients
  // (B as stored, transposing reads) that the streaming and phased kernels do not take follow the same rule: FastVim-B
  // in_proj data gradient (N = 768, K = 3072) 3.0 -> 2.5, the channel model's (M = 100 352) 6.0 -> 5.0 / 10.5 -> 10.0
  static const int m160 = fv_tune("FASTVIM_GEMM_M160", 1);   // tuning hook
  if (m160 && !tall && AMODE == KC && whole_k && splits == 1 && p.N % 128 == 0 && p.M >= 4096) {
    const long slots = 2L * fv_cu_count();
    auto cost = [&](long bm, long bn) { return fv_cdiv((long)fv_cdiv(p.M, bm) * (p.N / bn), slots) * bm * bn; };
    const bool wide_ok = p.N % 192 == 0 && p.N >= 384 && !(p.N == 768 && p.K <= 192);
    const bool n96_ok = p.N % 96 == 0 && p.N < 384;
    // against the shape that would run otherwise (128 x 192 where the width allows it), with 8 % in hand for the taller
    // tile's larger L2 -> LDS fill per flop
    const long cur = wide_ok ? cost(128, 192) : cost(128, 128);
    if (!n96_ok && cost(160, 128) * 27 / 25 < cur) return launch_k<AMODE, BMODE, 2, 2, true, 4, 5>(p, splits, st);
  }
  // N = 192 (FastVim-T: out_proj forward, in_proj data gradient, patch embed): two 96-wide tiles cover it exactly,
  // two 128-wide ones compute and load a quarter too much
  static const int n96 = fv_tune("FASTVIM_GEMM_N96", 1);   // tuning hook (out_proj forward 11.2 -> 10.1 us, in_proj dgrad 18.0 -> 16.9)
  if (n96 && !tall && AMODE == KC && p.N % 96 == 0 && p.N < 384 && whole_k)
    return launch_k<AMODE, BMODE, 2, 2, true, 3>(p, splits, st);
  static const bool wide = (fv_tune("FASTVIM_GEMM_N192", 1) != 0);   // tuning hook
  if (wide && AMODE == KC && p.N % 192 == 0 && p.N >= 384 && !(p.N == 768 && p.K <= 192) && p.K % BK == 0 &&
      p.k_per_split % BK == 0)
    return launch_k<AMODE, BMODE, 2, 2, true, 6>(p, splits, st);
  // 128x128 everywhere else: with LDS-DMA staging the 256x64 shape no longer pays at N = 192 (measured equal or
  // up to 8 % slower); it stays available for tuning
#ifdef FASTVIM_TUNING_HOOKS
  static const int force = fv_tune("FASTVIM_GEMM_TILE", 0);   // 41 = 256x64
  if (force == 41 && p.M >= 256) return launch<AMODE, BMODE, 4, 1>(p, splits, st);
#endif
  return launch<AMODE, BMODE, 2, 2>(p, splits, st);
}

}  // namespace

static int gemm_entry(const void* A, const void* B, void* C, const float* bias, int rb_period, int M, int N, int K,
                      long lda, long ldb, long ldc, int a_k_slow, int b_k_slow, int c_fp32, int splits,
                      fv_stream_t stream);

extern "C" int fv_gemm_bf16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K,
                            long lda, long ldb, long ldc, int a_k_slow, int b_k_slow, int c_fp32, int splits,
                            fv_stream_t stream) {
  return gemm_entry(A, B, C, bias, 0, M, N, K, lda, ldb, ldc, a_k_slow, b_k_slow, c_fp32, splits, stream);
}

extern "C" int fv_gemm_bf16_rowbias(const void* A, const void* B, float* C, const float* table, int period, int M, int N,
                                    int K, long lda, long ldb, long ldc, fv_stream_t stream) {
  FV_CHECK(table && period > 0, "gemm_bf16_rowbias: needs a (period, N) table");
  return gemm_entry(A, B, C, table, period, M, N, K, lda, ldb, ldc, 0, 0, 1, 1, stream);
}

static int gemm_entry(const void* A, const void* B, void* C, const float* bias, int rb_period, int M, int N, int K,
                      long lda, long ldb, long ldc, int a_k_slow, int b_k_slow, int c_fp32, int splits,
                      fv_stream_t stream) {
  FV_CHECK(A && B && C, "gemm_bf16: null pointer");
  FV_CHECK(M > 0 && N > 0 && K > 0 && splits >= 1, "gemm_bf16: empty problem");
  FV_CHECK(lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16: leading dimensions must be multiples of 8 (16-byte rows)");
  FV_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm_bf16: operands must be 16-byte aligned");
  FV_CHECK(ldc % 4 == 0 && ((uintptr_t)C & 15) == 0, "gemm_bf16: C must be 16-byte aligned with ldc %% 4 == 0");
  FV_CHECK(splits == 1 || c_fp32, "gemm_bf16: split-K partials must be fp32");
  FV_CHECK(a_k_slow ? M % 8 == 0 : K % 8 == 0, "gemm_bf16: A's contiguous extent must be a multiple of 8");
  FV_CHECK(b_k_slow ? N % 8 == 0 : K % 8 == 0, "gemm_bf16: B's contiguous extent must be a multiple of 8");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.c_fp32 = c_fp32; p.rb_period = rb_period;
  int kps = fv_cdiv(fv_cdiv(K, splits), BK) * BK;
  p.k_per_split = kps;
  p.c_split_stride = (long)M * ldc;
  const int zs = fv_cdiv(K, kps);
  FV_CHECK(zs == splits, "gemm_bf16: K=%d cannot be cut into %d slices of whole 64-deep tiles (got %d)", K, splits, zs);
  hipStream_t st = (hipStream_t)stream;
  if (!a_k_slow && !b_k_slow) return launch_shape<KC, KC>(p, splits, st);
  if (!a_k_slow && b_k_slow) return launch_shape<KC, KS>(p, splits, st);
  if (a_k_slow && b_k_slow) return launch_shape<KS, KS>(p, splits, st);
  fv_set_error("gemm_bf16: A K-slow with B K-contiguous is not built");
  return FV_ERR_UNSUPPORTED;
}

// Rows per workgroup of the fused projection + norm kernels (64-row MFMA tiles, two workgroups resident per CU).  Dealing
// the rows out evenly over whole rounds of the 2 x CUs slots -- M = 25 088 (FastVim-T, batch 128) as 512 x 49 instead of
// 392 x 64, where 136 CUs carry two tiles and 120 carry one -- was measured and LOST: every workgroup streams the whole
// weight panel (442 KB for the backward kernel) from L2 whatever its height, so more, shorter workgroups cost more than
// the imbalance (same box, FastVim-T step: 64 rows 5.822 ms, 56 rows 5.805-5.816, 49 rows 5.844-5.865, 33 rows 6.41).
// The full 64 rows stay; FASTVIM_FUSED_RPT = n (tuning build) sets another height, 0 the even deal.
static int fused_rpt(int M) {
  static const int force = fv_tune("FASTVIM_FUSED_RPT", 64);      // tuning hook
  if (force > 0) return force < 64 ? force : 64;
  const int cus = fv_cu_count();
  const long slots = 2L * cus, rounds = (M + 64 * slots - 1) / (64 * slots);
  long rpt = (M + slots * rounds - 1) / (slots * rounds);
  if (rpt < 16) rpt = 16;
  return (int)(rpt > 64 ? 64 : rpt);
}

extern "C" int fv_gemm_bf16_addnorm(const void* A, const void* W, const float* residual, const float* norm_weight,
                                    const float* row_scale, int rows_per_scale, void* y, float* residual_out, float* rstd,
                                    int M, int N, int K, long lda, long ldw, float eps, fv_stream_t stream) {
  return fv_gemm_bf16_addnorm2(A, W, residual, norm_weight, row_scale, rows_per_scale, y, residual_out, rstd, M, N, K, lda, ldw, eps,
                               nullptr, nullptr, 0, 0, stream);
}

extern "C" int fv_gemm_bf16_addnorm2(const void* A, const void* W, const float* residual, const float* norm_weight,
                                     const float* row_scale, int rows_per_scale, void* y, float* residual_out, float* rstd,
                                     int M, int N, int K, long lda, long ldw, float eps, const void* W2, void* C2, int N2,
                                     long ldw2, fv_stream_t stream) {
  FV_CHECK(A && W && residual && norm_weight && y && residual_out && rstd, "gemm_bf16_addnorm: null pointer");
  FV_CHECK(M > 0 && K > 0, "gemm_bf16_addnorm: empty problem");
  if (N != 192 || K % BK != 0) return FV_ERR_UNSUPPORTED;      // whole 192-wide rows per workgroup, LDS-DMA staging
  FV_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
               ((uintptr_t)residual & 15) == 0 && ((uintptr_t)residual_out & 15) == 0 && ((uintptr_t)y & 7) == 0,
           "gemm_bf16_addnorm: operands must be 16-byte aligned with row strides multiples of 8");
  FV_CHECK(!row_scale || rows_per_scale > 0, "gemm_bf16_addnorm: rows_per_scale must be positive");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = y; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = N; p.c_fp32 = 0;
  p.k_per_split = K;
  if (W2) {
    FV_CHECK(C2 && N2 > 0 && N2 % 128 == 0 && ldw2 >= N && ldw2 % 8 == 0 && ((uintptr_t)W2 & 15) == 0 && ((uintptr_t)C2 & 15) == 0,
             "gemm_bf16_addnorm2: the second weight must be (N2, N) with N2 a multiple of 128, 16-byte aligned");
  }
  NormEpi ne{residual, norm_weight, row_scale, residual_out, (bf16_t*)y, rstd, rows_per_scale > 0 ? rows_per_scale : 1, eps,
             nullptr, nullptr, (const bf16_t*)W2, (bf16_t*)C2, ldw2, N2, fused_rpt(M)};
#ifdef FASTVIM_TUNING_HOOKS
  ne.stamps = g_fv_stamps;
#endif
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto bm) {
    constexpr int BMR = decltype(bm)::value;
    // main loop: two (BMR + 192) x 64 stages; second phase: the normalised tile, two 128 x 64 stages, four epilogue slabs
    constexpr size_t s1 = (size_t)2 * (BMR + 192) * BK * 2;
    constexpr size_t s2 = ((size_t)BMR * 400 + 255) / 256 * 256 + 2 * 128 * BK * 2 + 4 * 32 * 144;
    const size_t smem = s1 > s2 ? s1 : s2;
    static FvOncePerDevice attr_set;   
    if (attr_set.first()) {
      (void)hipFuncSetAttribute((const void*)gemm_addnorm_kernel<BMR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)0;         
    }
    if (ne.rpt > BMR) ne.rpt = BMR;
    hipLaunchKernelGGL((gemm_addnorm_kernel<BMR>), dim3(fv_cdiv(M, ne.rpt)), dim3(256), smem, st, p, ne);
  };
#ifdef FASTVIM_TUNING_HOOKS
  static const int bm = fv_tune("FASTVIM_ADDNORM_BM", 64);
  if (bm == 32) go(std::integral_constant<int, 32>{});
  else if (bm == 128) go(std::integral_constant<int, 128>{});
  else
#endif
  go(std::integral_constant<int, 64>{});
  FV_LAUNCH_CHECK();
  return FV_OK;
}

#ifdef FASTVIM_TUNING_HOOKS
// diagnostic builds only: (workgroups, 8) uint64 device buffer the fused kernels stamp their phases into (NULL: off)
extern "C" void fv_debug_set_stamps(unsigned long long* buf) { g_fv_stamps = buf; }
extern "C" unsigned long long* fv_debug_get_stamps() { return g_fv_stamps; }      // for the other translation units' probes
#endif

extern "C" int fv_gemm_bf16_dgrad_addnorm_blocks(int M) { return fv_cdiv(M, fused_rpt(M)); }

extern "C" int fv_gemm_bf16_dgrad_addnorm_bwd(const void* A, const void* W, const float* dresidual_out, const float* r,
                                              const float* rstd, const float* norm_weight, const float* row_scale,
                                              int rows_per_scale, void* dx, float* dresidual_in, float* partial_dw, int M,
                                              int N, int K, long lda, long ldw, fv_stream_t stream) {
  return fv_gemm_bf16_dgrad_addnorm_bwd2(A, W, dresidual_out, r, rstd, norm_weight, row_scale, rows_per_scale, dx, dresidual_in,
                                         partial_dw, M, N, K, lda, ldw, nullptr, nullptr, 0, 0, stream);
}

extern "C" int fv_gemm_bf16_dgrad_addnorm_bwd2(const void* A, const void* W, const float* dresidual_out, const float* r,
                                               const float* rstd, const float* norm_weight, const float* row_scale,
                                               int rows_per_scale, void* dx, float* dresidual_in, float* partial_dw, int M,
                                               int N, int K, long lda, long ldw, const void* W2, void* C2, int N2, long ldw2,
                                               fv_stream_t stream) {
  FV_CHECK(A && W && r && rstd && norm_weight && dx && dresidual_in && partial_dw, "gemm_bf16_dgrad_addnorm_bwd: null pointer");
  FV_CHECK(M > 0 && K > 0, "gemm_bf16_dgrad_addnorm_bwd: empty problem");
  if (N != 192 || K % BK != 0) return FV_ERR_UNSUPPORTED;
  FV_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ldw >= N && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
               ((uintptr_t)r & 15) == 0 && ((uintptr_t)dresidual_in & 15) == 0 && ((uintptr_t)dx & 7) == 0 &&
               (!dresidual_out || ((uintptr_t)dresidual_out & 15) == 0),
           "gemm_bf16_dgrad_addnorm_bwd: operands must be 16-byte aligned with row strides multiples of 8");
  FV_CHECK(!row_scale || rows_per_scale > 0, "gemm_bf16_dgrad_addnorm_bwd: rows_per_scale must be positive");
  GemmParams p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)W; p.C = dx; p.bias = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = N; p.c_fp32 = 0;
  p.k_per_split = K;
  if (W2) {
    FV_CHECK(C2 && N2 > 0 && N2 % 128 == 0 && ldw2 >= N2 && ldw2 % 8 == 0 && ((uintptr_t)W2 & 15) == 0 && ((uintptr_t)C2 & 15) == 0,
             "gemm_bf16_dgrad_addnorm_bwd2: the second weight must be (N, N2) with N2 a multiple of 128, 16-byte aligned");
  }
  NormEpi ne{r, norm_weight, row_scale, dresidual_in, (bf16_t*)dx, const_cast<float*>(rstd),
             rows_per_scale > 0 ? rows_per_scale : 1, 0.f, dresidual_out, partial_dw,
             (const bf16_t*)W2, (bf16_t*)C2, ldw2, N2, fused_rpt(M)};
#ifdef FASTVIM_TUNING_HOOKS
  ne.stamps = g_fv_stamps;
#endif
  auto go = [&](auto bm) {
    constexpr int BMR = decltype(bm)::value;
    // main loop: two (BMR + 192) x 64 stages; second phase: the d x tile, two 128 x 64 stages, four epilogue slabs
    constexpr size_t s1 = (size_t)2 * (BMR + 192) * BK * 2;
    constexpr size_t s2 = ((size_t)BMR * 400 + 255) / 256 * 256 + 2 * 128 * BK * 2 + 4 * 32 * 144;
    const size_t smem = s1 > s2 ? s1 : s2;
    static FvOncePerDevice attr_set;   
    if (attr_set.first()) {
      (void)hipFuncSetAttribute((const void*)gemm_dgrad_addnorm_bwd_kernel<BMR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      (void)0;         
    }
    hipLaunchKernelGGL((gemm_dgrad_addnorm_bwd_kernel<BMR>), dim3(fv_cdiv(M, ne.rpt)), dim3(256), smem, (hipStream_t)stream, p, ne);
  };
  go(std::integral_constant<int, 64>{});      // (128-row tiles, no room to prefetch the rows: 46.9 vs 31.0 us)
  FV_LAUNCH_CHECK();
  return FV_OK;
}

// Grouped weight gradients: problem i is x_i (Kd_i, M_i)^T @ y_i (Kd_i, N_i) -> parts_i (splits_i, M_i, N_i) fp32,
// both operands K-slow (rows = tokens), split-K over whole 64-deep tiles -- the arithmetic and fixed split order of
// fv_gemm_bf16(a_k_slow = b_k_slow = 1, c_fp32 = 1), which launches them one at a time.
// Tile class of an S-width weight gradient (d_model 384: in_proj 1536 x 384 -> 4 = 256 x 192 tiles, out_proj 384 x 768 -> 5 =
// 192 x 256 tiles, eight waves, one workgroup per CU), 0 = not taken.  The eight-wave tiles pay from long K loops on
// (FastChannelVim-S, 100 352 tokens: step 44.35 -> 43.51 ms; FastVim-S at 224 px, 25 088 tokens: 12.55 -> 12.62, stays on
// 128 x 128; profiles/r05_ab_wgrad_s_width_tiles.log).  The host (fastvim_amd/gemm.py: _tile_class, split-K factors) and
// the launcher below both ask THIS function.
extern "C" int fv_gemm_bf16_tn_grouped_wide8(int M, int N, int Kd) {
  static const int wide8 = fv_tune("FASTVIM_WGRAD_WIDE8", 1);   // tuning hook
  if (!wide8 || Kd < 50000) return 0;
  if (M % 256 == 0 && N % 192 == 0 && N % 256 != 0 && (long)M * N >= 1024 * 384) return 4;
  if (M % 192 == 0 && N % 256 == 0 && M % 256 != 0 && (long)M * N >= 384 * 768) return 5;
  return 0;
}

extern "C" int fv_gemm_bf16_tn_grouped(const void* const* x, const void* const* y, float* const* parts, const int* Kd,
                                       const int* M, const int* N, const int* splits, int count, fv_stream_t stream) {
  return fv_gemm_bf16_tn_grouped_ld(x, y, parts, Kd, M, N, nullptr, nullptr, splits, count, stream);
}

extern "C" int fv_gemm_bf16_tn_grouped_ld(const void* const* x, const void* const* y, float* const* parts,
                                          const int* Kd, const int* M, const int* N, const int* ldx, const int* ldy,
                                          const int* splits, int count, fv_stream_t stream) {
  FV_CHECK(x && y && parts && Kd && M && N && splits && count > 0, "gemm_bf16_tn_grouped: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int launches = fv_cdiv(count, GROUP_MAX), per_launch = fv_cdiv(count, launches);   // balanced, no one-problem tail launch
  for (int base = 0; base < count; base += per_launch) {
    GroupedParams G{};
    const int n = count - base < per_launch ? count - base : per_launch;
    bool any_direct = false;
    for (int i = 0; i < n; ++i) {
      const int q = base + i;
      // splits[q] == -1: ONE K slice, ADDED to the (M, N) fp32 matrix parts[q] points at (large outputs of the phased
      // 256 x 256 form only): no partial buffer, no reduction launch
      const bool direct = splits[q] == -1;
      const int nsl = direct ? 1 : splits[q];
      any_direct = any_direct || direct;
      FV_CHECK(x[q] && y[q] && parts[q] && Kd[q] > 0 && M[q] > 0 && N[q] > 0 && nsl >= 1,
               "gemm_bf16_tn_grouped: bad problem %d", q);
      const int la = ldx ? ldx[q] : M[q], lb = ldy ? ldy[q] : N[q];
      FV_CHECK(la >= M[q] && lb >= N[q] && la % 8 == 0 && lb % 8 == 0 && ((uintptr_t)x[q] & 15) == 0 &&
                   ((uintptr_t)y[q] & 15) == 0 && ((uintptr_t)parts[q] & 15) == 0 && N[q] % 4 == 0,
               "gemm_bf16_tn_grouped: problem %d: operands must be 16-byte aligned, row strides multiples of 8", q);
      GemmParams& p = G.p[i];
      p.A = (const bf16_t*)x[q]; p.B = (const bf16_t*)y[q]; p.C = parts[q]; p.bias = nullptr;
      p.M = M[q]; p.N = N[q]; p.K = Kd[q]; p.lda = la; p.ldb = lb; p.ldc = N[q]; p.c_fp32 = direct ? 2 : 1;
      p.k_per_split = fv_cdiv(fv_cdiv(Kd[q], nsl), BK) * BK;
      p.c_split_stride = (long)M[q] * N[q];
      FV_CHECK(fv_cdiv(Kd[q], p.k_per_split) == nsl,
               "gemm_bf16_tn_grouped: problem %d: K=%d cannot be cut into %d slices of whole 64-deep tiles", q, Kd[q], nsl);
    }
    G.count = n;
    static const int xcd_order = fv_tune("FASTVIM_WGRAD_GROUP_XCD", 1) | (fv_tune("FASTVIM_GEMM_DBG", 0) << 8);   // tuning hooks
    // LDS-DMA staging needs whole 64-deep K tiles in every slice and whole 16-byte column groups
    static const bool ks_glds = (fv_tune("FASTVIM_WGRAD_GLDS", 1) != 0);   // tuning hook
    bool dma = ks_glds;
    for (int i = 0; i < n; ++i) {
      const GemmParams& q = G.p[i];
      dma = dma && q.K % BK == 0 && q.k_per_split % BK == 0 && ((q.M + 7) & ~7) <= q.lda && ((q.N + 7) & ~7) <= q.ldb &&
            q.M >= 8 && q.N >= 8;
    }
    // Tile shape of the launch: the one that pads the problems least -- 128 x 192 for outputs 192 wide (in_proj at
    // FastVim-T: 768 x 192), 192 x 128 for outputs 192 high (out_proj: 192 x 384), 128 x 128 otherwise and on ties.  A
    // 128 x 128 tiling computes a quarter more than the output there and re-reads the operands from L2 2.8x / 2.3x;
    // the launch is bound by L2 -> LDS traffic (phase probe: loads alone 194 us, multiply alone 133 us of 211).  The
    // caller groups problems of one shape class per call (gemm.py).
    static const int tile_env = fv_tune("FASTVIM_WGRAD_GROUP_TILE", 0);   // tuning hook: 1 = 128 x 128 always
    long area[3] = {0, 0, 0};
    const int bm[3] = {128, 128, 192}, bn[3] = {128, 192, 128};
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < 3; ++c)
        area[c] += (long)fv_cdiv(G.p[i].M, bm[c]) * bm[c] * fv_cdiv(G.p[i].N, bn[c]) * bn[c] * fv_cdiv(G.p[i].K, G.p[i].k_per_split);
    int cls = 0;
    if (dma && tile_env != 1) {
      if (area[1] < area[cls]) cls = 1;
      if (area[2] < area[cls]) cls = 2;
    }
    // 256 x 256 tiles on 8 waves when every output of the launch is a large multiple of them (FastVim-B: 3072 x 768 and
    // 768 x 1536; gemm.py puts those in a launch of their own): half the L2 -> LDS traffic of 128 x 128, weight-gradient
    // tail 5.6 -> 5.3 ms per FastVim-B step
    {
      bool big = dma && tile_env != 1;
      for (int i = 0; i < n; ++i)
        big = big && G.p[i].M % 256 == 0 && G.p[i].N % 256 == 0 && (long)G.p[i].M * G.p[i].N >= 512 * 512;
      if (big) {
        int b2 = 0;
        for (int i = 0; i < n; ++i) {
          b2 += (G.p[i].M / 256) * (G.p[i].N / 256) * fv_cdiv(G.p[i].K, G.p[i].k_per_split);
          G.blk_end[i] = b2;
        }
        static FvOncePerDevice attr;   
        if (attr.first()) {
          (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<KS, KS, 2, 4, true, 4, 8>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 256) * BK * 2);
          (void)hipFuncSetAttribute((const void*)gemm_p256_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * (256 + 256) * BK * 2);
          (void)0;     
        }
        static const int wg_phased = fv_tune("FASTVIM_WGRAD_P256", 1);   // tuning hook
        bool two_tiles = true;             // the phased loop wants at least two K tiles per slice
        for (int i = 0; i < n; ++i) two_tiles = two_tiles && G.p[i].k_per_split >= 2 * BK && G.p[i].K % G.p[i].k_per_split == 0;
        FV_CHECK(!any_direct || (wg_phased && two_tiles), "gemm_bf16_tn_grouped: in-place accumulation (splits = -1) needs the phased 256 x 256 form");
        if (wg_phased && two_tiles)
          hipLaunchKernelGGL(gemm_p256_grouped_kernel, dim3(b2), dim3(512), (size_t)2 * (256 + 256) * BK * 2, st, G, xcd_order);
        else
        hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 4, true, 4, 8>), dim3(b2), dim3(512),
                           (size_t)2 * (256 + 256) * BK * 2, st, G, xcd_order);
        FV_LAUNCH_CHECK();
        continue;
      }
    }
    FV_CHECK(!any_direct, "gemm_bf16_tn_grouped: in-place accumulation (splits = -1) is built for outputs that are multiples of 256 x 256 (>= 512 x 512) only");
    // S-width outputs (d_model 384: in_proj 1536 x 384, out_proj 384 x 768 -- FastVim-S, FastChannelVim-S): 256 x 192 /
    // 192 x 256 tiles on 8 waves, one workgroup per CU.  The launch is bound by the L2 -> LDS fill (see above): a 128 x 128
    // tile moves 32 KB per 2.1 MFLOP of K step, these 56 KB per 6.3 -- 0.59 x the fill (gemm.py puts the two shapes in
    // launches of their own)
    {
      // ONE rule decides (fv_gemm_bf16_tn_grouped_wide8: the host groups problems and picks split-K factors by it too --
      // round 5 had the K threshold on the Python side only, so a launch Python had classed for 128 x 128 tiles could run
      // on these with the wrong split factors)
      bool c4 = dma && tile_env != 1, c5 = c4;
      for (int i = 0; i < n; ++i) {
        const int w8 = fv_gemm_bf16_tn_grouped_wide8(G.p[i].M, G.p[i].N, G.p[i].K);
        c4 = c4 && w8 == 4;
        c5 = c5 && w8 == 5;
      }
      if (c4 || c5) {
        const int tm = c4 ? 256 : 192, tn = c4 ? 192 : 256;
        int b2 = 0;
        for (int i = 0; i < n; ++i) {
          b2 += (G.p[i].M / tm) * (G.p[i].N / tn) * fv_cdiv(G.p[i].K, G.p[i].k_per_split);
          G.blk_end[i] = b2;
        }
        const size_t sm8 = (size_t)2 * (256 + 192) * BK * 2;
        static FvOncePerDevice attr8;
        if (attr8.first()) {
          (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<KS, KS, 4, 2, true, 6, 4>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
          (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<KS, KS, 2, 4, true, 4, 6>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8);
        }
        if (c4) hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 4, 2, true, 6, 4>), dim3(b2), dim3(512), sm8, st, G, xcd_order);
        else hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 4, true, 4, 6>), dim3(b2), dim3(512), sm8, st, G, xcd_order);
        FV_LAUNCH_CHECK();
        continue;
      }
    }
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
      blocks += fv_cdiv(G.p[i].M, bm[cls]) * fv_cdiv(G.p[i].N, bn[cls]) * fv_cdiv(G.p[i].K, G.p[i].k_per_split);
      G.blk_end[i] = blocks;
    }
    const size_t smem = (size_t)2 * (bm[cls] + bn[cls]) * BK * 2;
    // (256 x 192 tiles on 8 waves for the first class, half the L2 traffic again: 310 vs 316 us in the step in round 2 -- not
    //  taken; again in round 3 with 6 / 8 / 12 / 14 K slices: step 5.73-5.84 against 5.68 ms, profiles/r03_ab_wgrad_t256_tiles.log)
    if (cls == 1) {
      static FvOncePerDevice attr;   
      if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<KS, KS, 2, 2, true, 6, 4>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)0;     
      }
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 2, true, 6, 4>), dim3(blocks), dim3(256), smem, st, G, xcd_order);
    } else if (cls == 2) {
      static FvOncePerDevice attr;   
      if (attr.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<KS, KS, 2, 2, true, 4, 6>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)0;     
      }
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 2, true, 4, 6>), dim3(blocks), dim3(256), smem, st, G, xcd_order);
    } else if (dma) {
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 2, true, 4, 4>), dim3(blocks), dim3(256), smem, st, G, xcd_order);
    } else {
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<KS, KS, 2, 2, false, 4, 4>), dim3(blocks), dim3(256), smem, st, G, xcd_order);
    }
    FV_LAUNCH_CHECK();
  }
  return FV_OK;
}
