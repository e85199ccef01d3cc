// out_proj + the next block's DropPath scale / residual add / RMSNorm with the WEIGHT HELD IN REGISTERS
// (mamba_simple_faster.py:435-444 + models/fastvim.py:168-190; same values as fv_gemm_bf16_addnorm, bit for bit).
//
// fv_gemm_bf16_addnorm (csrc/gemm_mfma.hip) tiles the M = 25 088 rows of a FastVim-T block into 392 workgroups of 64 rows,
// two per CU, and each of them streams the whole 192 x 384 weight (147 KB) through LDS for its 64 rows: 58 MB of weight
// fill per launch beside 19 MB of activations -- the launch is bound by bytes through the CUs' load path, not by HBM
// (phase stamps).  Here a workgroup is PERSISTENT, one per CU, 12 waves: wave w owns the 16 output
// columns [16 w, 16 w + 16) and keeps its K x 16 slice of the weight in registers for the whole launch (K = 384: 48
// VGPRs, loaded once, straight from global memory in MFMA-operand layout -- the weight is K-contiguous as stored); the
// rows of the workgroup (M / 256 = 98) pass through in tiles of 32: the activation tile is staged through registers into a
// double-buffered LDS image that all 12 waves read their A fragments from, the bf16-rounded product goes through an LDS
// tile into the norm epilogue (lane mapping and operation order of add_norm_fwd3_kernel<16>, as in the kernel this
// replaces), the next tile's rows and residual rows are in flight under the current tile's arithmetic.  Weight bytes per
// launch: 256 x 147 KB = 38 MB once, and none of it through LDS.
//
// MEASURED (round 4, profiles/r04_ab_addnorm_register_weights.log): bit-identical to the tiled kernel and exactly as
// fast -- 21.8 vs 21.3 us HBM-cold (3.1 vs 3.2 TB/s on the 67 MB both move), 18.8 vs 16.0 us cache-warm, FastVim-T step
// 5.72-5.73 vs 5.71-5.73 ms.  The weight fill was not what bounds this launch: two designs with 58 MB and 0 MB of weight
// traffic through LDS land on the same 21 us, which is what ANY launch moving 67 MB once gets on this chip (a plain 3-stream
// add of that size: 3.0-4.5 TB/s, bench.py roofline.elementwise_floor_same_size_cold).  Kept as a tested opt-in
// (mamba_simple_faster.ADDNORM_RW, bench.py --rw); the tiled kernel stays the default.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 rw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float rw_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rw_u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 12, NTHR = 64 * NW, BN = 192;   // 12 waves x 16 columns = whole 192-wide rows
constexpr int TR = 32;                             // rows per tile (two 16-row MFMA sub-tiles)
constexpr int PSB = BN * 2 + 16;                   // bytes per row of the product tile (bank spread)

struct RwParams {
  const bf16_t* A;          // (M, K) activations, row stride lda
  const bf16_t* W;          // (192, K) weight as stored (K-contiguous), row stride ldw
  const float* residual;    // (M, 192) fp32
  const float* w;           // (192) RMSNorm weight
  const float* row_scale;   // per-sample DropPath scale of the GEMM output, nullable
  float* res_out;           // (M, 192) fp32
  bf16_t* y;                // (M, 192) normalised rows
  float* rstd;              // (M)
  long lda, ldw;
  int M, rows_per_scale, rows_per_wg;
  float eps;
};

template <int KSTEPS>      // K / 32
__global__ __launch_bounds__(NTHR) void gemm_addnorm_rw_kernel(RwParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = 32 * KSTEPS, AS = K * 2 + 16;           // bytes per staged activation row (+16: bank spread)
  constexpr int VPR = K / 8, NV = TR * VPR / NTHR;          // 16-byte vectors per row / per thread and tile
  static_assert(TR * VPR % NTHR == 0, "the tile must deal out evenly");
  char* sP = smem + 2 * TR * AS;      // [A image 0 | A image 1 | product tile]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r0 = blockIdx.x * p.rows_per_wg, r1 = min(p.M, r0 + p.rows_per_wg);
  if (r0 >= r1) return;
  const int ntile = (r1 - r0 + TR - 1) / TR;
  const int lr = lane & 15, gr = lane >> 4;
  // ---- this wave's slice of the weight: rows n = 16 wv + (lane & 15), 8 consecutive k per lane and 32-deep step
  rw_bf16x8 wf[KSTEPS];
  {
    const bf16_t* wr = p.W + (size_t)(16 * wv + lr) * p.ldw + gr * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) wf[j] = *reinterpret_cast<const rw_bf16x8*>(wr + 32 * j);
  }
  // ---- staging of an activation tile through registers
  rw_u32x4 av[NV];
  auto a_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NTHR, row = e / VPR, v = e - row * VPR;
      const int gr_ = r0 + t * TR + row;
      const rw_u32x4 z = {0u, 0u, 0u, 0u};
      av[i] = gr_ < r1 ? *reinterpret_cast<const rw_u32x4*>(p.A + (size_t)gr_ * p.lda + v * 8) : z;
    }
  };
  auto a_store = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NTHR, row = e / VPR, v = e - row * VPR;
      *reinterpret_cast<rw_u32x4*>(buf + row * AS + v * 16) = av[i];
    }
  };
  // ---- epilogue rows of a tile: waves 0 .. 7 take 4 rows each (16 lanes per row, 3 x 4 channels per lane)
  const bool epi = wv < TR / 4;
  float4 rr[3];
  float sc_next = 1.f;
  auto r_load = [&](int t) {
    if (epi) {
      const int row = r0 + t * TR + wv * 4 + gr, rowc = row < r1 ? row : r1 - 1;
#pragma unroll
      for (int k = 0; k < 3; ++k) rr[k] = *reinterpret_cast<const float4*>(p.residual + (size_t)rowc * BN + (k * 16 + lr) * 4);
      sc_next = p.row_scale ? p.row_scale[rowc / p.rows_per_scale] : 1.f;
    }
  };
  float w[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(p.w + (k * 16 + lr) * 4);
    w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
  }
  const float inv_n = 1.f / (float)BN;

  a_load(0);
  r_load(0);
  a_store(smem);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const char* cur = smem + (t & 1) * (TR * AS);
    float4 rc[3] = {rr[0], rr[1], rr[2]};
    const float sc = sc_next;
    if (t + 1 < ntile) {       // the next tile's rows leave for the registers now and land under this tile's arithmetic
      a_load(t + 1);
      r_load(t + 1);
    }
    // ---- product of the tile: two 16-row sub-tiles x this wave's 16 columns, k in order (one accumulation chain each)
    rw_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const rw_bf16x8 a = *reinterpret_cast<const rw_bf16x8*>(cur + (s * 16 + lr) * AS + (32 * j + gr * 8) * 2);
        acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a, acc[s], 0, 0, 0);     // acc[s][i] = C[row lr][col 16 wv + 4 gr + i]
      }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint2 pk = {pack_bf16x2(acc[s][0], acc[s][1]), pack_bf16x2(acc[s][2], acc[s][3])};
      *reinterpret_cast<uint2*>(sP + (s * 16 + lr) * PSB + (16 * wv + 4 * gr) * 2) = pk;
    }
    __syncthreads();
    // ---- residual add + RMSNorm of the tile's rows (fv_add_norm_fwd on the bf16 product)
    if (epi) {
      const int rl = wv * 4 + gr, row = r0 + t * TR + rl;
      const bool live = row < r1;
      const size_t base = (size_t)(live ? row : r1 - 1) * BN;
      float v[3][4];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * 16 + lr) * 4;
        const uint2 xb = *reinterpret_cast<const uint2*>(sP + rl * PSB + c * 2);
        v[k][0] = __uint_as_float(xb.x << 16); v[k][1] = __uint_as_float(xb.x & 0xffff0000u);
        v[k][2] = __uint_as_float(xb.y << 16); v[k][3] = __uint_as_float(xb.y & 0xffff0000u);
        const float r4[4] = {rc[k].x, rc[k].y, rc[k].z, rc[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[k][e] = fmaf(v[k][e], sc, r4[e]);
        if (live) *reinterpret_cast<float4*>(p.res_out + base + c) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
      }
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) q = fmaf(v[k][e], v[k][e], q);
#define RW_DPP_ADD(ctrl) q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), ctrl, 0xf, 0xf, true))
      RW_DPP_ADD(0xB1);
      RW_DPP_ADD(0x4E);
      RW_DPP_ADD(0x141);
      RW_DPP_ADD(0x140);
#undef RW_DPP_ADD
      const float rstd = rsqrtf(q * inv_n + p.eps);
      if (lr == 0 && live) p.rstd[row] = rstd;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int c = (k * 16 + lr) * 4;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float tt = v[k][e] * rstd;          // (r * rstd) * w, pinned: the order add_norm_fwd3_kernel uses
          asm volatile("" : "+v"(tt));
          o[e] = tt * w[k][e];
        }
        const uint2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        if (live) *reinterpret_cast<uint2*>(p.y + base + c) = pk;
      }
    }
    if (t + 1 < ntile) a_store(smem + ((t + 1) & 1) * (TR * AS));
    __syncthreads();          // the next tile's image is complete; the product tile may be overwritten
  }
}

}  // namespace

extern "C" int fv_gemm_bf16_addnorm_rw_ok(int M, int N, int K) {
  static const int off = (fv_tune("FASTVIM_ADDNORM_RW", 1) == 0);     // A/B hook
  return !off && N == BN && (K == 192 || K == 384) && M >= 1;      // (K = 768 would spill: 96 weight registers)
}

extern "C" int fv_gemm_bf16_addnorm_rw(const void* A, const void* W, const float* residual, const float* norm_weight,
                                       const float* row_scale, int rows_per_scale, void* y, float* residual_out,
                                       float* rstd, int M, int N, int K, long lda, long ldw, float eps,
                                       fv_stream_t stream) {
  FV_CHECK(A && W && residual && norm_weight && y && residual_out && rstd, "gemm_bf16_addnorm_rw: null pointer");
  FV_CHECK(fv_gemm_bf16_addnorm_rw_ok(M, N, K), "gemm_bf16_addnorm_rw: built for N = 192, K in {192, 384} (got N %d, K %d)", N, K);
  FV_CHECK(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
               ((uintptr_t)residual & 15) == 0 && ((uintptr_t)residual_out & 15) == 0 && ((uintptr_t)y & 7) == 0 &&
               ((uintptr_t)norm_weight & 15) == 0,
           "gemm_bf16_addnorm_rw: operands must be 16-byte aligned with row strides multiples of 8");
  FV_CHECK(!row_scale || rows_per_scale > 0, "gemm_bf16_addnorm_rw: rows_per_scale must be positive");
  RwParams p{};
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.residual = residual; p.w = norm_weight; p.row_scale = row_scale;
  p.res_out = residual_out; p.y = (bf16_t*)y; p.rstd = rstd; p.lda = lda; p.ldw = ldw; p.M = M;
  p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; p.eps = eps;
  // one workgroup per CU, the rows dealt out evenly (M = 25 088 on 256 CUs: 98 each)
  const int cus = fv_cu_count();
  int grid = fv_cdiv(M, TR) < cus ? fv_cdiv(M, TR) : cus;
  p.rows_per_wg = fv_cdiv(M, grid);
  grid = fv_cdiv(M, p.rows_per_wg);
  hipStream_t st = (hipStream_t)stream;
#define FV_RW(KS_)                                                                                             \
  do {                                                                                                         \
    const size_t smem = (size_t)2 * TR * (32 * KS_ * 2 + 16) + (size_t)TR * PSB;                               \
    static FvOncePerDevice done;                                                                               \
    if (smem > 64 * 1024 && done.first())                                                                      \
      (void)hipFuncSetAttribute((const void*)gemm_addnorm_rw_kernel<KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)smem);                                                                    \
    hipLaunchKernelGGL((gemm_addnorm_rw_kernel<KS_>), dim3(grid), dim3(NTHR), smem, st, p);                    \
  } while (0)
  if (K == 192) FV_RW(6); else FV_RW(12);
#undef FV_RW
  FV_LAUNCH_CHECK();
  return FV_OK;
}
