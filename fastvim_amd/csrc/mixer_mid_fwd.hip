// hipcc-flags: -fno-slp-vectorize
// EXPERIMENT (round 4, measured, NOT on the default path): the middle of a FastVim mixer's forward pass in ONE launch --
// everything between in_proj and out_proj (mamba_simple_faster.py:272-444):
//     conv1d + SiLU of both directions, pooling over the grid columns, skip = D conv_f + D_b conv_b      (272-305, 356-358)
//     x_proj, dt_proj + softplus, the selective scan over the pooled rows, per direction                  (321-354, 384-410)
//     expand + skip + direction average + LayerNorm + SiLU(z) gate                                        (356, 412-441)
// -- for the 14 x 14 / 16 x 16 grids of the 224 / 256 px models at d_inner = 384, bf16.  It replaces the three launches
// conv_pool_fwd_row -> xproj_scan_fwd_short -> combine_fwd_wave with the same per-lane arithmetic and the same summation
// orders (outputs agree to the last bit except where -ffast-math contracts the two translation units differently: a few
// elements per million one bf16 ulp apart, tests/test_mixer_mid_gpu.py).
//
// Decomposition.  Everything in that range is independent per image, and there are 128 images for 256 CUs: an image is
// carried by a PAIR of workgroups (one per CU, 12 waves), split a different way in each phase so that no phase does
// redundant work and every exchange is a tensor the backward pass needs in HBM anyway:
//   P1  conv + pool + skip     split by pooling ROWS   (half h owns rows [h R/2, (h+1) R/2): row-local work)
//   X1  each half publishes its rows of the pooled tensor xc (both directions)
//   P2  x_proj + dt_proj + scan split by DIRECTION     (half h scans direction h over all R rows, all channels)
//   X2  each half publishes the scan output yc of its direction
//   P3  combine                split by ROWS again     (LayerNorm statistics are token-local)
// The hand-offs follow MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility", table row
// "ONE lane of each storing workgroup": every handed-off byte is stored sc1 (write-through, agent scope), every storing
// wave drains (s_waitcnt vmcnt(0)), the workgroup barrier, ONE lane stores the flag sc1; the consumer polls the flag with
// one lane (sc1 loads, s_sleep), joins a workgroup barrier, and reads the bytes with sc1 loads only.  The consumer resets
// the flag it has consumed, so a flag buffer is all zero again when the launch ends.  Both halves of a pair are resident
// together because the launch is at most one workgroup per CU (the host refuses larger batches; LDS use keeps it at
// one per CU); every spin is bounded and reports through an error word instead of hanging.
//
// What it measured (FastVim-T 224 px, batch 128, same box, profiles/r04_ab_mid_fwd_fusion.log, phase stamps in
// profiles/r04_mid_fwd_phase_stamps_v*.log): the launch takes 45-47 us against 44 us for the three launches inside the
// step; the step 5.75-5.78 ms either way.  A workgroup lives 40 us: conv + pool 5-6, the two hand-offs 6.5 + 5.5 (publish,
// partner skew, flag round trip) plus 1.5 + 2 for reading what the partner published, x_proj 3, the two scan chunks 4.5
// each, combine 7-10.  Fusion removes two launch boundaries and the skip round trip through HBM, and pays them back in
// hand-offs: with one half-image per CU there is no second image on a CU whose memory phase could run under this one's
// scan -- the per-CU chain conv -> scan -> combine is as long as before.  Kept as a tested opt-in
// (fastvim_amd.mixer_ops.MID_FWD, bench.py --mid-fusion); the three launches stay the default.
#include <stdlib.h>

#include "lane_reduce.h"
#include "mixer_common.h"
#include "packed.h"

namespace {

constexpr int N = 16;                      // d_state
constexpr int NWV = 12, NTHR = 64 * NWV;   // one workgroup per CU: 12 waves, 3 per SIMD
constexpr int CH = 16 * NWV;               // channels of a scan chunk (a wave owns 16)
constexpr int SPIN_MAX = 1 << 21;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 mm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sf2 sfma2(sf2 a, sf2 b, sf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ sf2 sexp2_2(sf2 a) { sf2 o; o.x = fv_exp2(a.x); o.y = fv_exp2(a.y); return o; }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }

struct MidParams {
  const void* xz;                        // (B, L, 2 d_in) bf16
  const float *wf, *bf, *wb, *bb;        // conv1d / conv1d_b (d_in, 4), (d_in)
  const float *Df, *Db;                  // (d_in)
  const void* Wx2;                       // (2, R + 2N, d_in) bf16
  const float* Wdt[2];                   // (d_in, R)
  const float* dtb[2];                   // (d_in)
  const float* Alog[2];                  // (d_in, N)
  const float *lnw, *lnb;                // (d_in) or null
  void* xc;                              // out (2, B, rows, d_in) bf16
  void* skip;                            // out (B, L, d_in) bf16
  void* xdbl;                            // out (2, B rows, R + 2N) bf16
  float* yc;                             // out (2, B, rows, d_in) fp32
  void* g;                               // out (B, L, d_in) bf16
  float *mean, *rstd;                    // out (B L)
  int* flags;                            // (4 B + 1): [2 b + h] xc published, [2 B + 2 b + h] yc published, [4 B] error word
  Geo geo;
  int B, d_in, R;
  float pool_scale, eps;
  int use_norm;
  unsigned long long* stamps;            // tuning builds: phase stamps (fv_debug_set_stamps), else null
};

// agent-scope (sc1) dword accesses: the only way handed-off bytes are touched
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// publish: every wave has drained its stores, the workgroup has met, one lane raises this half's flag.
// await: one lane polls the partner's flag and takes it down again, the workgroup joins.  Work that does not depend on the
// partner (stores the partner does not read, loads of inputs) is issued BETWEEN the two: it then runs under the wait instead
// of in front of the flag.
__device__ __forceinline__ void publish(int* mine) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) st_sc1((uint32_t*)mine, 1u);
}
__device__ __forceinline__ void await(int* theirs, int* err) {
  if (threadIdx.x == 0) {
    int n = 0;
    while (ld_sc1((const uint32_t*)theirs) == 0u) {
      __builtin_amdgcn_s_sleep(2);
      if (++n > SPIN_MAX) {
        st_sc1((uint32_t*)err, 1u);
        break;
      }
    }
    st_sc1((uint32_t*)theirs, 0u);
  }
  __syncthreads();
}

template <int RQ, int LCT>
struct MidLds {          // byte offsets; the scan part is xproj_scan_fwd_short's layout
  static constexpr int RQP = (RQ + 3) / 4 * 4, WP = 4 * RQP + 2 * N;
  static constexpr int xc_stride(int d_in) { return d_in * 2 + 16; }
  static constexpr int o_xc = 0;
  static constexpr int o_dbl(int d_in) { return o_xc + 16 * xc_stride(d_in); }
  static constexpr int o_ch(int d_in) { return o_dbl(d_in) + LCT * WP * 4; }
  static constexpr int o_y(int d_in) { return o_ch(d_in) + LCT * CH * 8; }
  static constexpr int o_xp(int d_in) { return o_y(d_in) + LCT * CH * 4; }
  static constexpr int o_ys(int d_in, int NTW) { return o_xp(d_in) + NWV * NTW * 256 * 4; }      // (LCT / 2) x d_in fp32
  static constexpr int bytes(int d_in, int NTW) { return o_ys(d_in, NTW) + (LCT / 2) * d_in * 4; }
};

#ifdef FASTVIM_TUNING_HOOKS
extern "C" unsigned long long* fv_debug_get_stamps();
// phase stamps (diagnostic build; tools/probe/mid_stamps.py): thread 0 stores s_memtime into [workgroup][16]
__device__ __forceinline__ void mid_stamp(unsigned long long* buf, int slot) {
  if (buf && threadIdx.x == 0) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    buf[(size_t)blockIdx.x * 16 + slot] = t;
  }
}
#define MID_STAMP(slot) mid_stamp(p.stamps, slot)
#else
#define MID_STAMP(slot) ((void)0)
#endif

// RQ: quads of dt_rank; LCT: pooling rows = scan length (14 or 16, even); NT: grid columns (14 or 16)
template <int RQ, int LCT, int NT>
__global__ __launch_bounds__(NTHR) void mixer_mid_fwd_kernel(MidParams p, int NTW) {
  extern __shared__ __attribute__((aligned(16))) char smc[];
  typedef MidLds<RQ, LCT> LD;
  typedef PairVec<bf16_t, 1> P1;
  typedef PairVec<bf16_t, 3> P3;
  constexpr int RQP = LD::RQP, WP = LD::WP, R2 = LCT / 2, Lc = LCT;
  const int d_in = p.d_in, W = p.R + 2 * N;
  const int XS = LD::xc_stride(d_in);
  char* s_xc = smc + LD::o_xc;
  float* s_dbl = (float*)(smc + LD::o_dbl(d_in));
  float* s_ch = (float*)(smc + LD::o_ch(d_in));
  float* s_y = (float*)(smc + LD::o_y(d_in));
  float* s_xp = (float*)(smc + LD::o_xp(d_in));
  float* s_ys = (float*)(smc + LD::o_ys(d_in, NTW));
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x >> 1, h = blockIdx.x & 1;       // image, half: partners are neighbours in dispatch order
  const Geo g = p.geo;
  const int tok_x = 2 * d_in * 2, tok_s = d_in * 2;        // bytes per token of xz / of a (B, L, d_in) tensor
  const __amdgpu_buffer_rsrc_t bx = fv_make_buf((const bf16_t*)p.xz + (size_t)b * g.L * 2 * d_in, (size_t)g.L * tok_x);
  const __amdgpu_buffer_rsrc_t bs = fv_make_buf((bf16_t*)p.skip + (size_t)b * g.L * d_in, (size_t)g.L * tok_s);
  const size_t xdir = (size_t)p.B * Lc * d_in;             // elements between the directions of xc / yc
  uint32_t* xc_w = (uint32_t*)p.xc;                        // bf16 pairs

  MID_STAMP(0);
  // ------------------------------------------------------------------ P1: conv + pool + skip of rows [h R2, (h+1) R2)
  // (conv_pool_fwd_row_kernel<bf16, NT, 1, false>: a lane owns a channel pair of one row; 4 rows per round of 12 waves)
  {
    constexpr int WPR = 3, RPR = NWV / WPR, NRND = (R2 + RPR - 1) / RPR;       // d_in = 384: three waves per row
    const int grp = wv / WPR, c0 = ((wv - grp * WPR) * 64 + lane) * 2;
    const int voff = c0 * 2;
    f2 wf[CW], wb[CW];
    load_taps2(p.wf, c0, wf);
    load_taps2(p.wb, c0, wb);
    const f2 bf = load_f2(p.bf, c0), bb = load_f2(p.bb, c0);
    const f2 Df = load_f2(p.Df, c0), Db = load_f2(p.Db, c0);
    P1 xr[NRND][NT + 6];
    uint32_t skp[NRND][NT];                    // skip of this lane's tokens, packed: stored behind the flag (below)
#pragma unroll
    for (int r = 0; r < NRND; ++r) {           // every round's row in flight before the first SiLU
      const int li = r * RPR + grp;
      if (li < R2) {
        const int i = h * R2 + li, m_row = i * g.s_i;
        const int s_up = i > 0 ? -g.s_i : 0, s_dn = i + 1 < g.rows ? g.s_i : 0;
#pragma unroll
        for (int k = 0; k < NT + 6; ++k) {
          const int di = k < 3 ? -1 : (k >= NT + 3 ? 1 : 0);
          const int j = k - 3 - di * NT;
          xr[r][k].load(bx, voff, (m_row + (di < 0 ? s_up : di > 0 ? s_dn : 0) + j * g.s_j) * tok_x);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NRND; ++r) {
      const int li = r * RPR + grp;
      if (li < R2) {
        const int i = h * R2 + li, m_row = i * g.s_i;
        const float m_up = i > 0 ? 1.f : 0.f, m_dn = i + 1 < g.rows ? 1.f : 0.f;
        f2 accf = splat(0.f), accb = splat(0.f);
        f2 x[NT + 6];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = k < 3 ? xr[r][k].get(0) * m_up : xr[r][k].get(0);
#pragma unroll
        for (int jj = 0; jj < NT; ++jj) {
          x[jj + 6] = xr[r][jj + 6].get(0);
          if (jj + 3 >= NT) x[jj + 6] *= m_dn;
          f2 pf = bf, pb = bb;
#pragma unroll
          for (int k = 0; k < CW; ++k) {
            pf = fma2(wf[k], x[jj + k], pf);
            pb = fma2(wb[k], x[jj + 6 - k], pb);
          }
          const f2 xf = silu2(pf), xb = silu2(pb);
          const f2 sk = fma2(Df, xf, Db * xb);
          accf += xf;
          accb += xb;
          skp[r][jj] = pack_bf16x2(sk.x, sk.y);
        }
        const f2 a = accf * p.pool_scale, c = accb * p.pool_scale;
        const uint32_t pk[2] = {pack_bf16x2(a.x, a.y), pack_bf16x2(c.x, c.y)};
        const size_t o = (((size_t)b * Lc + i) * d_in + c0) >> 1;
        st_sc1(xc_w + o, pk[0]);                       // direction 0: half 1 scans it
        st_sc1(xc_w + (xdir >> 1) + o, pk[1]);         // direction 1: half 0
        const int srow = h ? Lc - 1 - i : i;           // my direction's rows go straight into the scan's staging area
        *reinterpret_cast<uint32_t*>(s_xc + srow * XS + c0 * 2) = pk[h];
      }
    }
    MID_STAMP(1);
    publish(p.flags + 2 * b + h);
    // skip is read by this workgroup only (P3) and by the backward pass: its 75 KB per workgroup leave under the wait
    // for the partner and under P2, not in front of the flag (a vmcnt(0) over them cost 4 us of every pair's hand-off)
#pragma unroll
    for (int r = 0; r < NRND; ++r) {
      const int li = r * RPR + grp;
      if (li < R2) {
        const int m_row = (h * R2 + li) * g.s_i;
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
          __builtin_amdgcn_raw_buffer_store_b32(skp[r][jj], bs, voff, (m_row + jj * g.s_j) * tok_s, 0);
      }
    }
  }
  int* fl = p.flags;
  await(fl + 2 * b + (1 - h), fl + 4 * p.B);
  MID_STAMP(2);
  // ------------------------------------------------------------------ P2: x_proj + dt_proj + scan of direction h
  const int dir = h;
  const size_t bd = ((size_t)dir * p.B + b) * Lc;
  {
    // the partner's rows of direction h (sc1 loads), rows past Lc zero
    const int wpr = d_in / 2;
    const uint32_t* src = xc_w + ((bd * d_in) >> 1);
    for (int e = tid; e < R2 * wpr; e += NTHR) {
      const int li = e / wpr, w = e - li * wpr;
      const int i = (1 - h) * R2 + li;
      const int srow = dir ? Lc - 1 - i : i;
      *reinterpret_cast<uint32_t*>(s_xc + srow * XS + w * 4) = ld_sc1(src + (size_t)i * wpr + w);
    }
    for (int e = tid; e < (16 - Lc) * wpr; e += NTHR) {
      const int r = e / wpr, w = e - r * wpr;
      *reinterpret_cast<uint32_t*>(s_xc + (Lc + r) * XS + w * 4) = 0u;
    }
    for (int e = tid; e < LCT * WP; e += NTHR) s_dbl[e] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staging loads, and this wave's skip stores: visible to P3's loads
  __syncthreads();
  MID_STAMP(3);
  {      // x_proj: wave w takes the 32-deep K steps w, w + 12, ...
    f32x4_t acc[7];
#pragma unroll
    for (int nt = 0; nt < 7; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kc = lane >> 4;
    const bf16_t* Wd = (const bf16_t*)p.Wx2 + (size_t)dir * W * d_in;
    for (int ks = wv; ks < d_in / 32; ks += NWV) {
      const int k0 = ks * 32 + kc * 8;
      const mm_bf16x8 a = *reinterpret_cast<const mm_bf16x8*>(s_xc + r * XS + k0 * 2);
#pragma unroll
      for (int nt = 0; nt < 7; ++nt) {
        if (nt < NTW) {
          const int n = min(nt * 16 + r, W - 1);
          const mm_bf16x8 bw = *reinterpret_cast<const mm_bf16x8*>(Wd + (size_t)n * d_in + k0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, a, acc[nt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 7; ++nt)
      if (nt < NTW) *reinterpret_cast<f32x4_t*>(s_xp + ((size_t)(wv * NTW + nt) * 64 + lane) * 4) = acc[nt];
  }
  __syncthreads();
  MID_STAMP(4);
  {
    bf16_t* xo = (bf16_t*)p.xdbl + bd * W;
    for (int e = tid; e < Lc * W; e += NTHR) {
      const int t = e / W, n = e - t * W;
      const int nt = n >> 4, ln = t + 16 * ((n & 15) >> 2), j = n & 3;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) v += s_xp[((size_t)(w * NTW + nt) * 64 + ln) * 4 + j];
      const bf16_t vb = __float2bfloat16(v);
      const int l = dir ? Lc - 1 - t : t;
      xo[(size_t)l * W + n] = vb;
      const int pos = n < p.R ? (n & 3) * RQP + (n >> 2) : 4 * RQP + (n - p.R);
      s_dbl[t * WP + pos] = __bfloat162float(vb);
    }
  }
  __syncthreads();
  MID_STAMP(5);
  {
    const int q = lane & 3, cm = lane & 15, tg = lane >> 4;
    const float* my_bc = s_dbl + 4 * RQP + q * 4;
    uint32_t* yc_w = (uint32_t*)p.yc;
    for (int ch0 = 0; ch0 < d_in; ch0 += CH) {
      {
        const int dm = ch0 + wv * 16 + cm;
        const bool actm = dm < d_in;
        const int ddm = actm ? dm : 0;
        const float bias_m = p.dtb[dir][ddm];
        f32x4_t D = {0.f, 0.f, 0.f, 0.f};
        const int ta = min(cm, LCT - 1);
#pragma unroll
        for (int kg = 0; kg < RQP; ++kg) {
          const float a = s_dbl[ta * WP + tg * RQP + kg];
          const int r = 4 * kg + tg;
          const float w = (actm && r < p.R) ? p.Wdt[dir][(size_t)ddm * p.R + r] : 0.f;
          D = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w, D, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = 4 * tg + r;
          if (s < LCT) {
            const float dt = actm ? fv_softplus(D[r] + bias_m) : 0.f;
            const float u = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(s_xc + s * XS + ddm * 2));
            *reinterpret_cast<float2*>(s_ch + ((size_t)s * CH + wv * 16 + cm) * 2) = make_float2(dt, dt * u);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const int chl = wv * 16 + (lane >> 2), d = ch0 + chl;
        const int dd = d < d_in ? d : 0;
        sf2 A2[2], st[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          A2[hh].x = -__expf(p.Alog[dir][(size_t)dd * N + q * 4 + 2 * hh]) * FV_LOG2E;
          A2[hh].y = -__expf(p.Alog[dir][(size_t)dd * N + q * 4 + 2 * hh + 1]) * FV_LOG2E;
        }
        const float* my_ch = s_ch + (size_t)chl * 2;
#pragma unroll
        for (int s = 0; s < LCT; ++s) {
          asm volatile("" ::: "memory");
          const float4 Bv = *reinterpret_cast<const float4*>(my_bc + s * WP);
          const float4 Cv = *reinterpret_cast<const float4*>(my_bc + s * WP + N);
          const float2 cv = *reinterpret_cast<const float2*>(my_ch + s * (CH * 2));
          const sf2 Bn[2] = {{Bv.x, Bv.y}, {Bv.z, Bv.w}}, Cn[2] = {{Cv.x, Cv.y}, {Cv.z, Cv.w}};
          sf2 acc2 = {0.f, 0.f};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            st[hh] = sfma2(sexp2_2(A2[hh] * cv.x), st[hh], Bn[hh] * cv.y);
            acc2 = sfma2(Cn[hh], st[hh], acc2);
          }
          const float acc = quad_sum(acc2.x + acc2.y);
          if (q == 0) s_y[s * CH + chl] = acc;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const int dm = ch0 + wv * 16 + cm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s = 4 * tg + r;
          if (dm < d_in && s < Lc) {
            const int l = dir ? Lc - 1 - s : s;
            const float y = s_y[s * CH + wv * 16 + cm];
            st_sc1(yc_w + (bd + l) * d_in + dm, __float_as_uint(y));
            const int li = l - h * R2;
            if (li >= 0 && li < R2) s_ys[li * d_in + dm] = y;       // my rows: P3 reads them from here
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      MID_STAMP(6 + ch0 / CH);
    }
  }
  // ------------------------------------------------------------------ P3: combine of rows [h R2, (h+1) R2)
  // (combine_fwd_wave_kernel<bf16, 3, 1, 2>: a wave owns a token, a lane 6 channels; here two PAIRS of tokens in flight
  //  per wave, the first two requested before the wait for the partner's scan output)
  constexpr int TT = 2, NPAIR = R2 * NT / TT;
  const int lc = lane * 6, voff3 = lc * 2;
  const __amdgpu_buffer_rsrc_t bz = fv_make_buf((const bf16_t*)p.xz + (size_t)b * g.L * 2 * d_in + d_in, (size_t)g.L * tok_x - tok_s);
  const __amdgpu_buffer_rsrc_t bg = fv_make_buf((bf16_t*)p.g + (size_t)b * g.L * d_in, (size_t)g.L * tok_s);
  P3 nsk[2][TT], nz[2][TT];
  auto tok_of = [&](int pr, int t) {         // memory token of the t-th token of pair pr
    const int li = (pr * TT) / NT, j = pr * TT - li * NT + t;
    return (h * R2 + li) * g.s_i + j * g.s_j;
  };
  auto fetch = [&](int pr, int slot) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int m = tok_of(pr, t);
      nsk[slot][t].load(bs, voff3, m * tok_s);
      nz[slot][t].load(bz, voff3, m * tok_x);
    }
  };
  publish(fl + 2 * p.B + 2 * b + h);
  if (wv < NPAIR) fetch(wv, 0);
  if (wv + NWV < NPAIR) fetch(wv + NWV, 1);
  f2 lw[3], lb[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    lw[q] = p.use_norm ? load_f2(p.lnw, lc + 2 * q) : splat(1.f);
    lb[q] = p.use_norm ? load_f2(p.lnb, lc + 2 * q) : splat(0.f);
  }
  await(fl + 2 * p.B + 2 * b + (1 - h), fl + 4 * p.B);
  MID_STAMP(8);
  {
    // ys = yc_f + yc_b of my rows: the partner's direction through sc1 loads
    const uint32_t* src = (const uint32_t*)p.yc + ((size_t)(1 - h) * p.B + b) * Lc * d_in + (size_t)h * R2 * d_in;
    for (int e = tid; e < R2 * d_in; e += NTHR) {
      const float o = __uint_as_float(ld_sc1(src + e));
      s_ys[e] = h ? o + s_ys[e] : s_ys[e] + o;           // direction 0 first, as combine_fwd_wave adds them
    }
  }
  __syncthreads();
  MID_STAMP(9);
  {
    const float inv_d = 1.f / (float)d_in;
    int it = 0;
#pragma unroll 1
    for (int pr = wv; pr < NPAIR; pr += NWV, it ^= 1) {
      const int li = (pr * TT) / NT;
      f2 ys[3], o[TT][3], z[TT][3];
      float s1[TT], mean[TT], rstd[TT];
#pragma unroll
      for (int q = 0; q < 3; ++q) ys[q] = *reinterpret_cast<const f2*>(s_ys + li * d_in + lc + 2 * q);
      // (slot = it is a run-time index: both slots are unpacked with selects so the arrays stay in registers)
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f2 a0 = nsk[0][t].get(q), a1 = nsk[1][t].get(q), b0 = nz[0][t].get(q), b1 = nz[1][t].get(q);
          o[t][q] = it ? a1 : a0;
          z[t][q] = it ? b1 : b0;
        }
      if (pr + 2 * NWV < NPAIR) {
        if (it) fetch(pr + 2 * NWV, 1); else fetch(pr + 2 * NWV, 0);
      }
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        f2 acc = splat(0.f);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          o[t][q] = (ys[q] + o[t][q]) * 0.5f;
          acc += o[t][q];
        }
        s1[t] = hsum(acc);
      }
      if (p.use_norm) {
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
        const int m = tok_of(pr, t);
        f2 out[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) out[q] = fma2((o[t][q] - mean[t]) * rstd[t], lw[q], lb[q]) * silu2(z[t][q]);
        P3::store(bg, voff3, m * tok_s, out);
        if (p.use_norm && lane == 0) {
          p.mean[(size_t)b * g.L + m] = mean[t];
          p.rstd[(size_t)b * g.L + m] = rstd[t];
        }
      }
    }
  }
  MID_STAMP(10);
}

}  // namespace

// The shapes the fused launch is built for; everything else takes the three launches.
extern "C" int fv_mixer_mid_fwd_ok(int batch, int rows, int cols, int tpp, int d_inner, int dt_rank, int dtype, int pool_max) {
  static const int off = (fv_tune("FASTVIM_MID_FWD", 1) == 0);     // A/B hook
  if (off || dtype != FV_BF16 || pool_max || tpp != 1) return 0;
  if (!((rows == 14 && cols == 14) || (rows == 16 && cols == 16))) return 0;
  if (d_inner != 384 || dt_rank < 1 || dt_rank > 24) return 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return batch >= 1 && 2 * batch <= cus;       // both halves of every pair resident: at most one workgroup per CU
}

extern "C" int fv_mixer_mid_fwd(const void* xz, const float* conv_w, const float* conv_b, const float* conv_w_b,
                                const float* conv_b_b, const float* D, const float* D_b, const void* x_proj_w2,
                                const float* dt_w, const float* dt_bias, const float* A_log, const float* dt_w_b,
                                const float* dt_bias_b, const float* A_log_b, const float* ln_w, const float* ln_b,
                                void* xc, void* skip, void* x_dbl, float* yc, void* g, float* mean, float* rstd,
                                int* flags, int batch, int rows, int cols, int s_i, int s_j, int d_inner, int dt_rank,
                                int d_state, float scaling_factor, float eps, int dtype, fv_stream_t stream) {
  FV_CHECK(d_state == N, "mixer_mid_fwd: only d_state == 16 is built (got %d)", d_state);
  FV_CHECK(fv_mixer_mid_fwd_ok(batch, rows, cols, 1, d_inner, dt_rank, dtype, 0),
           "mixer_mid_fwd: shape not built (batch %d, grid %d x %d, d_inner %d, dt_rank %d): ask fv_mixer_mid_fwd_ok first",
           batch, rows, cols, d_inner, dt_rank);
  FV_CHECK(xz && conv_w && conv_w_b && D && D_b && x_proj_w2 && dt_w && dt_bias && A_log && dt_w_b && dt_bias_b && A_log_b &&
               xc && skip && x_dbl && yc && g && flags && (!ln_w || (mean && rstd)),
           "mixer_mid_fwd: null pointer");
  FV_CHECK(((uintptr_t)xz & 15) == 0 && ((uintptr_t)x_proj_w2 & 15) == 0 && ((uintptr_t)xc & 3) == 0,
           "mixer_mid_fwd: operands must be 16-byte aligned");
  MidParams p{};
  p.xz = xz; p.wf = conv_w; p.bf = conv_b; p.wb = conv_w_b; p.bb = conv_b_b; p.Df = D; p.Db = D_b;
  p.Wx2 = x_proj_w2;
  p.Wdt[0] = dt_w; p.Wdt[1] = dt_w_b; p.dtb[0] = dt_bias; p.dtb[1] = dt_bias_b; p.Alog[0] = A_log; p.Alog[1] = A_log_b;
  p.lnw = ln_w; p.lnb = ln_b;
  p.xc = xc; p.skip = skip; p.xdbl = x_dbl; p.yc = yc; p.g = g; p.mean = mean; p.rstd = rstd; p.flags = flags;
  p.geo = make_geo(rows, cols, s_i, s_j, 1);
  p.B = batch; p.d_in = d_inner; p.R = dt_rank;
  p.pool_scale = scaling_factor / (float)cols; p.eps = eps; p.use_norm = ln_w != nullptr;
#ifdef FASTVIM_TUNING_HOOKS
  p.stamps = fv_debug_get_stamps();      // csrc/gemm_mfma.hip, set by fv_debug_set_stamps()
#endif
  const int RQ = (dt_rank + 3) / 4, NTW = fv_cdiv(dt_rank + 2 * N, 16);
  dim3 grid(2 * batch), block(NTHR);
  hipStream_t st = (hipStream_t)stream;
#define FV_MID(RQQ, LCC)                                                                          \
  do {                                                                                            \
    /* at least 82 KiB: never two of these workgroups on one CU (the hand-offs are measured for one per CU) */ \
    size_t smem = (size_t)MidLds<RQQ, LCC>::bytes(d_inner, NTW);                                  \
    if (smem < 82 * 1024) smem = 82 * 1024;                                                       \
    static FvOncePerDevice done;                                                                  \
    if (done.first())                                                                             \
      (void)hipFuncSetAttribute((const void*)mixer_mid_fwd_kernel<RQQ, LCC, LCC>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);          \
    hipLaunchKernelGGL((mixer_mid_fwd_kernel<RQQ, LCC, LCC>), grid, block, smem, st, p, NTW);      \
  } while (0)
  if (rows == 14) { if (RQ <= 3) FV_MID(3, 14); else FV_MID(6, 14); }
  else { if (RQ <= 3) FV_MID(3, 16); else FV_MID(6, 16); }
#undef FV_MID
  FV_LAUNCH_CHECK();
  return FV_OK;
}
