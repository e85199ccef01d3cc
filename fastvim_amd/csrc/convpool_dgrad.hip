// hipcc-flags: -fno-slp-vectorize -fgpu-flush-denormals-to-zero
// conv + pool ADJOINT as the A-TILE PRODUCER of the in_proj data gradient whose epilogue is the block's residual add +
// RMSNorm adjoint, with the previous block's out_proj data gradient as a second GEMM phase --
// fv_mixer_conv_pool_bwd2 + fv_gemm_bf16_dgrad_addnorm_bwd2 in ONE launch, d_inner = 384, d_model = 192, bf16 (FastVim-T).
// The backward mirror of csrc/combine_gemm.hip.  Reference: FastVim_MambaInnerFnNoOutProj_withoutZ.backward
// (selective_scan_interface.py:607-776: causal_conv1d_bwd x 2, the mean-pool adjoint, `dxz @ in_proj.weight`),
// mamba_simple_faster.py:189-193, 270-305 and the fused add + norm of models/fastvim.py:168-190.
//
// Why the two fit without a hand-off: both are partitioned by tokens, and the conv's halo (3 tokens either side of a
// pooling row) is only READ.  A workgroup owns four pooling rows of one image (up to 64 tokens: tile row 16 w + j = column
// j of pooling row i0 + w), like the forward kernel;
//   phase 1  wave w runs conv_pool_bwd_row_kernel's arithmetic on pooling row i0 + w, one 128-channel chunk after the
//            other (a lane owns a channel pair; the next chunk's packed tokens are requested into the registers this chunk
//            has just unpacked), stores the x half of d xz to HBM -- the in_proj weight gradient needs it -- and into a
//            64 x 384 bf16 panel in LDS.  Per chunk the four waves' parameter-gradient sums meet in LDS in a fixed order
//            (two barriers) and leave as one partial row per workgroup;
//   phase 2  C (64 x 192) = [panel | z half of d xz] @ W_in: wave w owns output columns [48 w, 48 w + 48) and streams its
//            quarter of the TRANSPOSED weight (192 x 768, K-contiguous: a lane's MFMA fragment is 16 contiguous bytes)
//            from L2 straight into operand registers; the x half has no barrier and no A traffic, the z half -- written by
//            combine_bwd, read here for the only time -- comes through a three-stage LDS-DMA ring whose first stages
//            are requested before the K loop;
//   phase 3  gemm_dgrad_addnorm_bwd_kernel<64>'s epilogue from the product tile on, lane for lane (d hidden and d residual
//            bit-identical for the same d xz; the norm weight's row sums are grouped per pooling-row tile);
//   phase 4  d g of the previous block = d x tile @ W_out, tile_times_w2's code with the tile's row map.
// d xz's x half is read back by the weight gradient only: 19.3 MB of reads, one launch boundary and the cold start of a
// 392-workgroup GEMM per block go away.  LDS: 75 KB per workgroup, two workgroups per CU.
#include "mixer_common.h"
#include "packed.h"
#include "gemm_tiles.h"

namespace {

constexpr int CD_DI = 384, CD_N = 192, CD_K = 2 * CD_DI, CD_BM = 64, CD_NT = 256, CD_NCH = CD_DI / 128;
constexpr int CD_RSA = CD_DI * 2 + 16;         // x panel row stride (bytes): = 4 dwords (mod 64), fragment reads conflict-free
constexpr int CD_RSB = CD_N * 2 + 16;          // product tile row stride of the norm-adjoint epilogue (gemm_mfma.hip)
#ifndef CD_PDEPTH
#define CD_PDEPTH 4
#endif
#ifndef CD_SB_EVERY
#define CD_SB_EVERY 1    // conv steps per scheduling region (1: the stand-alone kernel's step-by-step order)
#endif
#ifndef CD_W2_PIPE
#define CD_W2_PIPE 1     // second phase: the next panel's first weight stage requested before this panel's epilogue
#endif
constexpr int CD_PD = CD_PDEPTH;               // k steps of weight fragments in flight per wave (refilled in line pairs)
constexpr int CD_KS = CD_K / 32;               // 24 k steps; 0..11 the x half (panel), 12..23 the z half (ring)
constexpr int CD_ZST = CD_BM * BK * 2;         // one z stage: 64 rows x 64 k, bf16, [row][64] with the KC chunk swizzle
constexpr int CD_NZ = CD_DI / BK;              // 6 z stages
constexpr int CD_O_RING = CD_BM * CD_RSA;      // 50 176: three z stages; phase 1: the waves' parameter-gradient slots
constexpr int CD_SLOT = 12 * 128;              // floats of one wave's parameter-gradient slot (one 128-channel chunk)
constexpr int CD_O_B = (CD_BM * CD_RSB + 255) / 256 * 256, CD_O_S = CD_O_B + 2 * 128 * BK * 2;     // second phase (gemm_mfma.hip)
constexpr int CD_SMEM = CD_O_S + 4 * 32 * (64 * 2 + 16);                                           // 76 800
static_assert(CD_O_RING + 3 * CD_ZST <= CD_SMEM && CD_O_RING + 4 * CD_SLOT * 4 <= CD_SMEM, "LDS regions");
static_assert(2 * CD_SMEM <= 160 * 1024, "two workgroups per CU");
#ifndef CD_DBG
#define CD_DBG 0      // phase probes (tuning): 1 no conv arithmetic, 2 no K loop, 3 no epilogue, 4 no second phase
#endif

struct CdParams {
  // ---- conv + pool adjoint (fvi::BwdParams of fv_mixer_conv_pool_bwd2)
  const void* xz;          // (B, L, 768) bf16: the x half is the conv input
  const void* dob;         // (B, L, 384) bf16: gradient of the skip tensor (from combine_bwd)
  const float* dxc;        // (2, B, rows, 384) fp32 pooled gradient
  const void* dxc2;        // nullable: second addend of the pooled gradient (bf16; fv_mixer_scan_bwd_xproj)
  const float *wf, *bf, *wb, *bb, *Df, *Db;
  void* dxz;               // (B, L, 768) bf16: x half WRITTEN here; z half (written by combine_bwd) READ here
  float* part;             // (workgroups, 12 * 384) parameter-gradient partial rows
  float pool_scale;
  Geo geo;
  int B;
  // ---- in_proj data gradient + norm adjoint (NormEpi of fv_gemm_bf16_dgrad_addnorm_bwd2)
  const bf16_t* Wt;        // (192, ldwt) bf16: in_proj.weight^T, K-contiguous
  long ldwt;
  const float* dres_out;   // (M, 192) fp32 gradient of the residual stream from above, nullable
  const float* r;          // (M, 192) fp32 saved normalisation input
  const float* rstd;       // (M)
  const float* nw;         // (192) RMSNorm weight
  const float* row_scale;  // DropPath scale per sample, nullable
  int rows_per_scale;
  bf16_t* dx;              // (M, 192) d hidden (x row_scale)
  float* dres_in;          // (M, 192)
  float* pw;               // (workgroups, 192) partial sums of d norm weight
  const bf16_t* W2;        // (192, ldw2): previous block's out_proj.weight as stored, nullable
  long ldw2;
  bf16_t* C2;              // (M, N2) d g of the previous block
  int N2;
  int M;
};

// LDS-DMA of 16 bytes per lane, issued as inline asm: through the builtin the compiler makes every LDS read that follows
// wait for ALL outstanding loads (it cannot tell the ring from the panel).  Completion is ours to wait for (vmcnt).
__device__ __forceinline__ void cd_dma16(const void* base, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// Memory operations a wave issues between the DMA of z stage t (t >= 3: behind the MFMAs of the iteration that read stage
// t - 3) and the wait for it at the top of stage t's iteration: per iteration in between, the weight refill (6 loads while
// one exists) and the next stage's DMA (2).
constexpr int cd_younger(int t) {
  constexpr int KX = CD_DI / 32;
  int n = 0;
  for (int q = KX + 2 * (t - 3) + 2; q < KX + 2 * t; q += 2) n += (q + CD_PD < CD_KS ? 6 : 0) + ((q - KX) / 2 + 3 < CD_NZ ? 2 : 0);
  return n;
}
static_assert(cd_younger(3) < 64 && cd_younger(4) < 64 && cd_younger(5) < 64, "vmcnt is a 6-bit counter");

template <int NT, bool X2>
__global__ __launch_bounds__(CD_NT, 2) void conv_pool_bwd_dgrad_kernel(CdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef PairVec<bf16_t, 1> P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Geo g = p.geo;
  const int tiles = (g.rows + 3) / 4;
  const int bimg = blockIdx.x / tiles, i0 = (blockIdx.x - bimg * tiles) * 4;
  const int mimg = bimg * g.L;                       // first memory token of the image
  auto tok_of = [&](int rl) {                        // tile row -> memory token, or -1
    const int w = rl >> 4, j = rl & 15;
    return (i0 + w < g.rows && j < g.cols) ? mimg + (i0 + w) * g.s_i + j * g.s_j : -1;
  };
  const int irow = i0 + wv;
  const bool wlive = irow < g.rows;                  // (a tile's last waves may have no pooling row)
  char* prow = smem + (wv * 16) * CD_RSA;            // this wave's 16 panel rows

  // ================= phase 1: conv + pool adjoint of pooling row irow, chunk by chunk =================
  {
    // panel rows that no token fills (columns >= cols; every row of a wave without a pooling row) must be finite
    const int zr0 = wlive ? NT : 0;
    for (int e = lane; e < (16 - zr0) * (CD_RSA / 4); e += 64) reinterpret_cast<uint32_t*>(prow + zr0 * CD_RSA)[e] = 0u;
    const int tok_x = 2 * CD_DI * 2, tok_d = CD_DI * 2;                   // bytes per token
    const size_t dstride = (size_t)p.B * g.rows * CD_DI;
    const __amdgpu_buffer_rsrc_t bp = fv_make_buf(p.dxc, 2 * dstride * 4);
    const __amdgpu_buffer_rsrc_t bp2 = fv_make_buf(X2 ? p.dxc2 : p.dxc, X2 ? 2 * dstride * 2 : 0);
    const int irc = wlive ? irow : 0;
    const bool up = irc > 0, down = irc + 1 < g.rows;
    const int m_row = irc * g.s_i, s_up = up ? -g.s_i : 0, s_dn = down ? g.s_i : 0;
    const int pooled_off = (bimg * g.rows + irc) * CD_DI;                // element offset of row irc's forward pooled gradient
    const size_t img_x = (size_t)bimg * g.L * 2 * CD_DI, img_d = (size_t)bimg * g.L * CD_DI;
    // a wave without a row gets zero-length descriptors: its loads return zeros and touch no memory
    const __amdgpu_buffer_rsrc_t bx = fv_make_buf((const bf16_t*)p.xz + img_x, wlive ? (size_t)g.L * tok_x : 0);
    const __amdgpu_buffer_rsrc_t bd = fv_make_buf((const bf16_t*)p.dob + img_d, wlive ? (size_t)g.L * tok_d : 0);
    const __amdgpu_buffer_rsrc_t bo = fv_make_buf((bf16_t*)p.dxz + img_x, wlive ? (size_t)g.L * tok_x : 0);
    struct Pooled {          // a pooled gradient as loaded: fp32 pair + (X2) the packed second addend
      f2 a;
      P b;
      __device__ __forceinline__ f2 sum() const { if constexpr (X2) return a + b.get(0); else return a; }
    };
    // pooled gradient of row irc - 1 + r (unscaled; chunk given by the channel offset c0)
    auto pooled = [&](int r, bool backward, int c0, bool on) {
      const bool ok = r == 1 || (r == 0 ? up : down);
      uint32_t w[2];
      const int eoff = pooled_off + (ok ? (r - 1) * CD_DI : 0) + (backward ? (int)dstride : 0);
      const int lane_off = on ? c0 : 0x1ffffff0;                            // (a chunk past the last: nothing is fetched)
      fv_buf_load_words<2>(bp, lane_off * 4, eoff * 4, w);
      Pooled o;
      o.a.x = __uint_as_float(w[0]);
      o.a.y = __uint_as_float(w[1]);
      if constexpr (X2) o.b.load(bp2, lane_off * 2, eoff * 2);
      return o;
    };
    P xr[NT + 6], dr[NT + 6];
    auto fetch = [&](int k, int voff) {
      const int di = k < 3 ? -1 : (k >= NT + 3 ? 1 : 0);
      const int j = k - 3 - di * NT;
      const int m = m_row + (di < 0 ? s_up : di > 0 ? s_dn : 0) + j * g.s_j;
      xr[k].load(bx, voff, m * tok_x);
      dr[k].load(bd, voff, m * tok_d);
    };
    struct Prm { f2 wf[CW], wb[CW], bf, bb, Df, Db; };
    auto load_prm = [&](int c0) {
      Prm q;
      load_taps2(p.wf, c0, q.wf);
      load_taps2(p.wb, c0, q.wb);
      q.bf = load_f2(p.bf ? p.bf : p.wf, c0);
      q.bb = load_f2(p.bb ? p.bb : p.wf, c0);
      q.Df = load_f2(p.Df ? p.Df : p.wf, c0);
      q.Db = load_f2(p.Db ? p.Db : p.wf, c0);
      return q;
    };
    // chunk 0: parameters first, the row right behind them, no wait in between (convpool_bwd_row.hip)
    Prm pn = load_prm(lane * 2);
    __builtin_amdgcn_sched_barrier(0);
    Pooled cf_mid_raw = pooled(1, false, lane * 2, true), cb_mid_raw = pooled(1, true, lane * 2, true);
    Pooled cb_up_raw = pooled(0, true, lane * 2, true);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NT + 6; ++k) fetch(k, lane * 4);
    __builtin_amdgcn_sched_barrier(0);
    Pooled cf_dn_raw = pooled(2, false, lane * 2, true);
    __builtin_amdgcn_sched_barrier(0);
    float* slot = reinterpret_cast<float*>(smem + CD_O_RING) + wv * CD_SLOT;
#pragma unroll 1
    for (int c = 0; c < CD_NCH; ++c) {
      const int c0 = c * 128 + lane * 2;                  // first channel of this lane's pair
      const int voff = c0 * 2;
      const bool more = c + 1 < CD_NCH;
      const int c0n = c0 + 128, voffn = more ? voff + 256 : 0x3ffffff0;     // next chunk (past the last: loads return zeros)
      f2 a_wf[CW], a_wb[CW], a_bf = splat(0.f), a_bb = splat(0.f), a_Df = splat(0.f), a_Db = splat(0.f);
#pragma unroll
      for (int k = 0; k < CW; ++k) a_wf[k] = a_wb[k] = splat(0.f);
      if (wlive && CD_DBG != 1) {          // uniform per wave
        f2 wf[CW], wb[CW];
#pragma unroll
        for (int k = 0; k < CW; ++k) { wf[k] = pn.wf[k]; wb[k] = pn.wb[k]; }
        const f2 bf = pn.bf * (p.bf ? 1.f : 0.f), bb = pn.bb * (p.bb ? 1.f : 0.f);
        const f2 Dfh = pn.Df * (p.Df ? 0.5f : 0.f), Dbh = pn.Db * (p.Db ? 0.5f : 0.f);
        const f2 cf_mid = cf_mid_raw.sum() * p.pool_scale, cb_mid = cb_mid_raw.sum() * p.pool_scale;
        cf_mid_raw = pooled(1, false, c0n, more);
        cb_mid_raw = pooled(1, true, c0n, more);
        const f2 cb_up = cb_up_raw.sum() * (up ? p.pool_scale : 0.f);
        f2 cf_dn = splat(0.f);
        f2 x[NT + 6], dov[NT + 6], dpf[NT + 6], dpb[NT + 6];   // index q + 3; live ranges are 4 steps (full unroll)
        const float m_up = up ? 1.f : 0.f, m_dn = down ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          x[k] = xr[k].get(0) * m_up;
          dov[k] = dr[k].get(0);
          dpf[k] = dpb[k] = splat(0.f);
          fetch(k, voffn);
        }
        if (more) pn = load_prm(c0n);         // (uniform; the next chunk's parameters arrive under this chunk's arithmetic)
#pragma unroll
        for (int n = -3; n < NT; ++n) {
          // step n: pre_f of position n+3 and pre_b of position n, both from x[n .. n+3]
          const int q3 = n + 3, k0 = n + 3, k3 = n + 6;             // array indices of positions n and n+3
          x[k3] = xr[k3].get(0);
          if (q3 >= NT) x[k3] *= m_dn;
          dov[k3] = dr[k3].get(0);
          fetch(k3, voffn);
          f2 pf = bf, pb = bb;
#pragma unroll
          for (int k = 0; k < CW; ++k) {
            pf = fma2(wf[k], x[k0 + k], pf);             // pre_f[n+3] = b + sum_k w[k] x[n+k]
            pb = fma2(wb[k], x[k3 - k], pb);             // pre_b[n]   = b + sum_k w[k] x[n+3-k]
          }
          const f2 sgf = sigmoid2(pf), sgb = sigmoid2(pb);
          const f2 dsf = sgf * fma2(pf, 1.f - sgf, splat(1.f)), dsb = sgb * fma2(pb, 1.f - sgb, splat(1.f));
          const float e3 = q3 < NT ? 1.f : m_dn, e0 = n >= 0 ? 1.f : m_up;
          if (q3 == NT) cf_dn = cf_dn_raw.sum() * (down ? p.pool_scale : 0.f);
          const f2 nf = fma2(Dfh, dov[k3], q3 >= NT ? cf_dn : cf_mid) * dsf * e3;
          const f2 nb = fma2(Dbh, dov[k0], n < 0 ? cb_up : cb_mid) * dsb * e0;
          if (n == -1) cb_up_raw = pooled(0, true, c0n, more);
          dpf[k3] = nf;
          dpb[k0] = nb;
          if (q3 < NT) {        // position n+3 belongs to this row: its parameter gradients are accumulated here
#pragma unroll
            for (int k = 0; k < CW; ++k) a_wf[k] = fma2(nf, x[k0 + k], a_wf[k]);
            a_bf += nf;
            a_Df = fma2(dov[k3], pf * sgf, a_Df);            // x 0.5 once, at the flush (exact)
          }
          if (n >= 0) {
#pragma unroll
            for (int k = 0; k < CW; ++k) a_wb[k] = fma2(nb, x[k3 - k], a_wb[k]);
            a_bb += nb;
            a_Db = fma2(dov[k0], pb * sgb, a_Db);
            // dx[n] = sum_k wf[k] dpre_f[n+3-k] + wb[k] dpre_b[n-3+k]
            f2 dx = splat(0.f);
#pragma unroll
            for (int k = 0; k < CW; ++k) {
              dx = fma2(wf[k], dpf[k3 - k], dx);
              dx = fma2(wb[k], dpb[k0 - 3 + k], dx);
            }
            const uint32_t pk[1] = {pack_bf16x2(dx.x, dx.y)};
            fv_buf_store_words<1>(bo, voff, (m_row + n * g.s_j) * tok_x, pk);
            *reinterpret_cast<uint32_t*>(prow + n * CD_RSA + voff) = pk[0];      // bank = 4 n + 64 c + lane: conflict-free
          }
          if ((n + 3) % CD_SB_EVERY == CD_SB_EVERY - 1) __builtin_amdgcn_sched_barrier(0);
        }
        cf_dn_raw = pooled(2, false, c0n, more);
      }
      // ---- the four waves' sums of this chunk meet in LDS (wave order = row order: fixed), one partial row per workgroup
      {
        const f2 hDf = a_Df * 0.5f, hDb = a_Db * 0.5f;
        float4* s4 = reinterpret_cast<float4*>(slot);
        s4[2 * lane] = make_float4(a_wf[0].x, a_wf[1].x, a_wf[2].x, a_wf[3].x);
        s4[2 * lane + 1] = make_float4(a_wf[0].y, a_wf[1].y, a_wf[2].y, a_wf[3].y);
        s4[128 + 2 * lane] = make_float4(a_wb[0].x, a_wb[1].x, a_wb[2].x, a_wb[3].x);
        s4[128 + 2 * lane + 1] = make_float4(a_wb[0].y, a_wb[1].y, a_wb[2].y, a_wb[3].y);
        f2* s2 = reinterpret_cast<f2*>(slot + 1024);
        s2[lane] = a_bf;
        s2[64 + lane] = a_bb;
        s2[128 + lane] = hDf;
        s2[192 + lane] = hDb;
      }
      __syncthreads();
      {
        const float* s0 = reinterpret_cast<const float*>(smem + CD_O_RING);
        float* dst = p.part + (size_t)blockIdx.x * 12 * CD_DI;
#pragma unroll
        for (int i = 0; i < CD_SLOT / CD_NT; ++i) {
          const int e = tid + i * CD_NT;
          const float v = ((s0[e] + s0[CD_SLOT + e]) + s0[2 * CD_SLOT + e]) + s0[3 * CD_SLOT + e];
          int gi;
          if (e < 512) gi = c * 512 + e;
          else if (e < 1024) gi = 4 * CD_DI + c * 512 + (e - 512);
          else {
            const int q = (e - 1024) >> 7, rr = (e - 1024) & 127;
            gi = (8 + q) * CD_DI + c * 128 + rr;
          }
          dst[gi] = v;
        }
      }
      __syncthreads();
    }
  }

  // ================= phase 2: C = [panel | z] @ W_in =================
  const int fn = lane & 15, fk = lane >> 4, lr = lane % 16, gr = lane / 16;
  const bf16_t* wrow[3];
  bf16x8 fb[CD_PD][3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) wrow[nb] = p.Wt + (long)(wv * 48 + nb * 16 + fn) * p.ldwt + fk * 8;
  // z ring: this thread's two 16-byte pieces of a stage (physical slot e = tid + 256 i -> row e >> 3, chunk (e & 7) ^ (row & 7))
  uint32_t zoff[2];
  const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(smem + CD_O_RING);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * CD_NT, rrow = e >> 3, ch = (e & 7) ^ (rrow & 7);
    const int m = tok_of(rrow);
    zoff[i] = (uint32_t)((m >= 0 ? m : mimg) * (CD_K * 2) + (CD_DI + ch * 8) * 2);
  }
  auto issue_z = [&](int t) {          // stage t -> ring buffer t % 3
#pragma unroll
    for (int i = 0; i < 2; ++i)
      cd_dma16(p.dxz, zoff[i] + (uint32_t)t * (BK * 2), ring + (uint32_t)(t % 3) * CD_ZST + (uint32_t)(i * CD_NT + wv * 64) * 16);
  };
  issue_z(0);
  issue_z(1);
  issue_z(2);
#pragma unroll
  for (int s = 0; s < CD_PD; s += 2)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int h = 0; h < 2; ++h) fb[s + h][nb] = *reinterpret_cast<const bf16x8*>(wrow[nb] + (s + h) * 32);
  __builtin_amdgcn_sched_barrier(0);
  // the saved normalisation rows of the epilogue (lane mapping of add_norm_bwd3_kernel<16>): requested here, they arrive
  // under the K loop
  float4 ne_r[2][2][3];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = tok_of(wv * 16 + it * 8 + u * 4 + gr), rowc = m >= 0 ? m : mimg;
#pragma unroll
      for (int k = 0; k < 3; ++k) ne_r[it][u][k] = *reinterpret_cast<const float4*>(p.r + (size_t)rowc * CD_N + (k * 16 + lr) * 4);
    }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();       // the panel is complete (and the parameter-gradient slots are free: the ring's DMA was issued after
                         // the last barrier of phase 1, which every wave passed after its last slot read)

  f32x4 acc[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* afrag = smem + (lane & 15) * CD_RSA + (lane >> 4) * 16;
  static_assert(CD_PD % 2 == 0 && CD_KS % 2 == 0, "the weight ring is refilled in line pairs");
  static_for<(CD_DBG == 2 ? 0 : CD_KS / 2)>([&](auto kp_c) {
    constexpr int kp = 2 * decltype(kp_c)::value;
    constexpr int KX = CD_DI / 32;                 // 12: first k step of the z half
    if constexpr (kp >= KX) {
      // ---- z half: stage t holds k steps kp, kp + 1.  Stages 0..2 were requested before the x half; stage t + 3 is
      //      requested into stage t's buffer once every wave has read it.
      //      Loads return in order, so "my pieces of stage t have landed" is a COUNTED wait: the operations this wave
      //      has issued behind DMA(t) may stay in flight -- the weight refills of the k-step pairs in between (6 loads each,
      //      while a refill exists: kp + CD_PD < CD_KS) and the two later stages' DMAs (2 each).  Stages 0..2 are older
      //      than weight fragments the x half has already consumed: no wait at all.  (vmcnt(0) here stalled every wave on
      //      its youngest weight refill, an L2 round trip per stage.)
      constexpr int t = (kp - KX) / 2;
      if constexpr (t >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cd_younger(t)) : "memory");
      if (t >= 3 || kp == KX) __builtin_amdgcn_s_barrier();      // ... and everyone's
    }
    bf16x8 cur[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) cur[h][nb] = fb[(kp + h) % CD_PD][nb];
    if constexpr (kp + CD_PD < CD_KS) {
#pragma unroll
      for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          fb[(kp + h) % CD_PD][nb] = *reinterpret_cast<const bf16x8*>(wrow[nb] + (kp + h + CD_PD) * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8 fa[4];
      if constexpr (kp < KX) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) fa[mb] = *reinterpret_cast<const bf16x8*>(afrag + mb * 16 * CD_RSA + (kp + h) * 64);
      } else {
        constexpr int t = (kp - KX) / 2;
        const char* st = smem + CD_O_RING + (t % 3) * CD_ZST;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) fa[mb] = frag<KC, CD_BM>(st, mb, h, lane);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[h][a], fa[b], acc[a][b], 0, 0, 0);
    }
    if constexpr (kp >= KX) {
      constexpr int t = (kp - KX) / 2;
      if constexpr (t + 3 < CD_NZ) {
        __builtin_amdgcn_s_barrier();        // every wave is done reading stage t
        issue_z(t + 3);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  // the residual-stream gradient rows of the epilogue: requested here, used at the end of each row's arithmetic
  float4 ne_g[2][2][3];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = tok_of(wv * 16 + it * 8 + u * 4 + gr), rowc = m >= 0 ? m : mimg;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        ne_g[it][u][k] = p.dres_out ? *reinterpret_cast<const float4*>(p.dres_out + (size_t)rowc * CD_N + (k * 16 + lr) * 4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  __syncthreads();       // every wave is done reading the panel and the ring: the product tile overlays them

  // ================= phase 3: RMSNorm + residual-add adjoint of the tile's rows =================
  //      gemm_bf16_body<..., NORM_EPI = 2> of gemm_mfma.hip from the product tile on, same lane mapping and operation order
  //      acc[a][b][j] = C[m = b * 16 + (lane & 15)][n = 48 wv + a * 16 + (lane >> 4) * 4 + j]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const f32x4 v = acc[a][b];
      uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *reinterpret_cast<uint2*>(smem + (b * 16 + (lane & 15)) * CD_RSB + (wv * 48 + a * 16 + (lane >> 4) * 4) * 2) = pk;
    }
  __syncthreads();
  if (CD_DBG == 3) return;
  {
    // (the norm kernels are built without the denormal flush and with source-order arithmetic where it matters: keep
    //  this block's contraction choices those of gemm_mfma.hip)
    constexpr int LPR = 16, RPW = 4, RU = 2, RW = CD_BM / 4;
    const float inv_n = 1.f / (float)CD_N;
    float w[3][4], aw[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(p.nw + (k * LPR + lr) * 4);
      w[k][0] = t.x; w[k][1] = t.y; w[k][2] = t.z; w[k][3] = t.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) aw[k][e] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < RW / (RPW * RU); ++it) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int rl = wv * RW + it * (RPW * RU) + u * RPW + gr;
        const int row = tok_of(rl);
        const bool live = row >= 0;
        const int rowc = live ? row : mimg;
        const size_t base = (size_t)rowc * CD_N;
        const float rstd = p.rstd[rowc];
        const float sc = p.row_scale ? p.row_scale[rowc / p.rows_per_scale] : 1.f;
        const float lv = live ? 1.f : 0.f;
        float xh[3][4], dxh[3][4];
        float c2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int c = (k * LPR + lr) * 4;
          const uint2 xb = *reinterpret_cast<const uint2*>(smem + rl * CD_RSB + c * 2);
          const float dyv[4] = {__uint_as_float(xb.x << 16), __uint_as_float(xb.x & 0xffff0000u),
                                __uint_as_float(xb.y << 16), __uint_as_float(xb.y & 0xffff0000u)};
          const float4 rr4 = ne_r[it][u][k];
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
          const float4 g4 = ne_g[it][u][k];
          const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
          float dr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) dr[e] = rstd * (dxh[k][e] - c1 - xh[k][e] * c2) + gg[e];
          if (live) {
            *reinterpret_cast<float4*>(p.dres_in + base + c) = make_float4(dr[0], dr[1], dr[2], dr[3]);
            if (p.row_scale)
#pragma unroll
              for (int e = 0; e < 4; ++e) dr[e] *= sc;
            uint2 pk = {pack_bf16x2(dr[0], dr[1]), pack_bf16x2(dr[2], dr[3])};
            *reinterpret_cast<uint2*>(p.dx + base + c) = pk;
            *reinterpret_cast<uint2*>(smem + rl * CD_RSB + c * 2) = pk;      // d x stays in the tile for the second phase
          } else {
            *reinterpret_cast<uint2*>(smem + rl * CD_RSB + c * 2) = make_uint2(0u, 0u);
          }
        }
      }
    }
    // d norm weight: the 4 row groups of a wave (permlane swaps), then the 4 waves through LDS, fixed order
    __syncthreads();
    float* s_acc = reinterpret_cast<float*>(smem + CD_BM * CD_RSB);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = aw[k][e];
        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
        auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
        if (gr == 0) s_acc[wv * CD_N + (k * LPR + lr) * 4 + e] = v;
      }
    __syncthreads();
    float* dst = p.pw + (size_t)blockIdx.x * CD_N;
    for (int c = tid; c < CD_N; c += CD_NT) dst[c] = (s_acc[c] + s_acc[CD_N + c]) + (s_acc[2 * CD_N + c] + s_acc[3 * CD_N + c]);
  }
  // ================= phase 4: the previous block's out_proj data gradient d g = d x @ W_out from the tile in LDS =================
  if (p.W2 && CD_DBG != 4) {
    __syncthreads();
    tile_times_w2_rows<CD_BM, CD_RSB, KS, decltype(tok_of), CD_W2_PIPE != 0>(smem, smem + CD_O_B, smem + CD_O_S, p.W2, p.ldw2, p.N2,
                                                                              p.C2, tok_of, tid);
  }
}

}  // namespace

extern "C" int fv_mixer_conv_pool_bwd_dgrad_ok(int batch, int rows, int cols, int tokens_per_patch, int d_inner, int d_model,
                                               int pool_max, int dtype) {
  return dtype == FV_BF16 && d_inner == CD_DI && d_model == CD_N && tokens_per_patch == 1 && !pool_max && batch > 0 &&
         rows > 0 && (cols == 14 || cols == 16) &&
         (long)batch * rows * cols * CD_K * 2 < 0x7fffffffL &&          // 32-bit byte offsets into d xz
         (long)batch * rows * CD_DI * 8 < 0x7ffff000L;                   // pooled gradients: one descriptor
}

extern "C" int fv_mixer_conv_pool_bwd_dgrad_blocks(int batch, int rows) { return batch * fv_cdiv(rows, 4); }

extern "C" int fv_mixer_conv_pool_bwd_dgrad(
    const void* xz, const void* dskip, const float* dxc, const void* dxc2, const float* conv_w, const float* conv_b,
    const float* conv_w_b, const float* conv_b_b, const float* D, const float* D_b, void* dxz, float* conv_partials, int batch,
    int rows, int cols, int tok_stride_row, int tok_stride_col, float scaling, const void* W_in_t, long ldwt,
    const float* dresidual_out, const float* r, const float* rstd, const float* norm_weight, const float* row_scale,
    int rows_per_scale, void* dx, float* dresidual_in, float* partial_dw, const void* W2, void* C2, int N2, long ldw2,
    fv_stream_t stream) {
  FV_CHECK(xz && dskip && dxc && conv_w && conv_w_b && dxz && conv_partials && W_in_t && r && rstd && norm_weight && dx &&
               dresidual_in && partial_dw, "mixer_conv_pool_bwd_dgrad: null pointer");
  FV_CHECK(fv_mixer_conv_pool_bwd_dgrad_ok(batch, rows, cols, 1, CD_DI, CD_N, 0, FV_BF16),
           "mixer_conv_pool_bwd_dgrad: shape not built (fv_mixer_conv_pool_bwd_dgrad_ok)");
  FV_CHECK((tok_stride_row == cols && tok_stride_col == 1) || (tok_stride_row == 1 && tok_stride_col == rows),
           "mixer_conv_pool_bwd_dgrad: token strides (%d,%d) are neither row-major nor transposed for a %dx%d grid",
           tok_stride_row, tok_stride_col, rows, cols);
  FV_CHECK(((uintptr_t)W_in_t & 15) == 0 && ldwt % 8 == 0 && ldwt >= CD_K, "mixer_conv_pool_bwd_dgrad: transposed weight rows must be 16-byte aligned");
  FV_CHECK(((uintptr_t)r & 15) == 0 && ((uintptr_t)dresidual_in & 15) == 0 && ((uintptr_t)dx & 7) == 0 &&
               ((uintptr_t)norm_weight & 15) == 0 && ((uintptr_t)dresidual_out & 15) == 0 && ((uintptr_t)dxz & 15) == 0,
           "mixer_conv_pool_bwd_dgrad: row operands must be 16-byte aligned");
  FV_CHECK(!row_scale || rows_per_scale > 0, "mixer_conv_pool_bwd_dgrad: rows_per_scale must be positive");
  FV_CHECK(!W2 || (C2 && N2 > 0 && N2 % 128 == 0 && ldw2 % 8 == 0 && ldw2 >= N2 && ((uintptr_t)W2 & 15) == 0 && ((uintptr_t)C2 & 15) == 0),
           "mixer_conv_pool_bwd_dgrad: the second weight must be (192, N2) with N2 a multiple of 128, 16-byte aligned");
  CdParams p{};
  p.xz = xz; p.dob = dskip; p.dxc = dxc; p.dxc2 = dxc2; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b;
  p.Df = D; p.Db = D_b; p.dxz = dxz; p.part = conv_partials;
  p.pool_scale = scaling / (float)cols;
  p.geo = make_geo(rows, cols, tok_stride_row, tok_stride_col, 1);
  p.B = batch;
  p.Wt = (const bf16_t*)W_in_t; p.ldwt = ldwt; p.dres_out = dresidual_out; p.r = r; p.rstd = rstd; p.nw = norm_weight;
  p.row_scale = row_scale; p.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; p.dx = (bf16_t*)dx;
  p.dres_in = dresidual_in; p.pw = partial_dw; p.W2 = (const bf16_t*)W2; p.ldw2 = ldw2; p.C2 = (bf16_t*)C2; p.N2 = N2;
  p.M = batch * rows * cols;
  const dim3 grid(batch * fv_cdiv(rows, 4)), block(CD_NT);
#define FV_CD(NTT, XX)                                                                                              \
  do {                                                                                                              \
    static FvOncePerDevice done;                                                                                    \
    if (done.first())                                                                                               \
      (void)hipFuncSetAttribute((const void*)conv_pool_bwd_dgrad_kernel<NTT, XX>, hipFuncAttributeMaxDynamicSharedMemorySize, CD_SMEM); \
    hipLaunchKernelGGL((conv_pool_bwd_dgrad_kernel<NTT, XX>), grid, block, CD_SMEM, (hipStream_t)stream, p);        \
  } while (0)
  if (cols == 14) { if (dxc2) FV_CD(14, true); else FV_CD(14, false); }
  else { if (dxc2) FV_CD(16, true); else FV_CD(16, false); }
#undef FV_CD
  FV_LAUNCH_CHECK();
  return FV_OK;
}
