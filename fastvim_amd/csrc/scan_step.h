// The packed adjoint step of the register-resident selective-scan backward kernels, shared by the fused mixer's pooled scan
// (csrc/scan_cl.hip: short and chunked kernels) and the reference-layout op's short-sequence backward (csrc/scan_bdl.hip):
// a lane = (channel, state quad) holds its four states as two packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32 do both halves
// in one issue slot); the 8 values of a step whose sums over the wave's 16 channels are d B / d C go through a reduce-scatter.
#pragma once
#include "common.h"
#include "lane_reduce.h"

typedef float sf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sf2 ssplat(float a) { sf2 o; o.x = a; o.y = a; return o; }
__device__ __forceinline__ sf2 sfma2(sf2 a, sf2 b, sf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ sf2 sexp2_2(sf2 a) { sf2 o; o.x = fv_exp2(a.x); o.y = fv_exp2(a.y); return o; }

// a * {b.lo, b.lo} and a * {b.lo, b.lo} + c on packed pairs
__device__ __forceinline__ sf2 pk_mul_lo(sf2 a, sf2 b) {
  sf2 o;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
__device__ __forceinline__ sf2 pk_fma_lo(sf2 a, sf2 b, sf2 c) {
  sf2 o;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(o) : "v"(a), "v"(b), "v"(c));
  return o;
}

// One step of the adjoint sweep of the register-resident backward kernels (short and chunked), for a lane = (channel, state
// quad) holding its four states as two packed pairs.  In: the step's B / C values of the quad, the table row {delta, u, dy,
// sigmoid}, the states after this step (xs) and before it (xp).  Out: the 8 values whose sums over the wave's 16 channels
// are d B / d C of the step, as pairs {dB01, dB23, dC01, dC23}; du_acc / ddraw for the channel.
//   a   = exp2(A2 delta)                    (re-derived: keeping it would cost a wave of occupancy)
//   dx  = C dy + dxa                        adjoint of the state after the step
//   dxa = a dx                              ... carried to the step before
//   pj  = dxa x_{t-1}                       (= dx a x_{t-1}: one product fewer than forming a x_{t-1} first)
struct AdjStep { float du_acc, ddraw; };
template <bool FIRST>
__device__ __forceinline__ AdjStep adjoint_step(const float4 Bv, const float4 Cv, const float4 cv, const sf2 (&A2)[2],
                                                const sf2 (&Araw)[2], const sf2 (&xs)[2], const sf2 (&xp)[2], sf2 (&dxa)[2],
                                                sf2 (&dA)[2], sf2 (&vals)[4]) {
  const sf2 Bn[2] = {{Bv.x, Bv.y}, {Bv.z, Bv.w}}, Cn[2] = {{Cv.x, Cv.y}, {Cv.z, Cv.w}};
  // the per-channel scalars multiply packed state pairs: as the LOW half of the register pair they were loaded into they
  // feed both halves of a packed op (op_sel_hi = 0) -- left to the compiler, the splats were built with v_mov pairs
  const sf2 ds = {cv.x, cv.y}, gs = {cv.z, cv.w};          // {delta, u}, {dy, sigmoid}
  const float uu = cv.y, sg = cv.w;
  sf2 tu;                                                   // {delta u, delta u}
  asm("v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(tu) : "v"(ds));
  sf2 du2 = {0.f, 0.f}, dd2 = {0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const sf2 a = sexp2_2(pk_mul_lo(A2[h], ds));
    const sf2 dx = pk_fma_lo(Cn[h], gs, dxa[h]);
    dxa[h] = a * dx;
    du2 = sfma2(dx, Bn[h], du2);
    if (!FIRST) {                                  // x_{-1} = 0 in the short kernel's first step
      const sf2 pj = dxa[h] * xp[h];
      dd2 = sfma2(Araw[h], pj, dd2);
      dA[h] = pk_fma_lo(pj, ds, dA[h]);
    }
    vals[h] = pk_mul_lo(dx, tu);                   // dB[4q + 2h ..]
    vals[2 + h] = pk_mul_lo(xs[h], gs);            // dC[4q + 2h ..]
  }
  AdjStep o;
  o.du_acc = quad_sum(du2.x + du2.y);
  const float dd_acc = quad_sum(dd2.x + dd2.y);
  // d delta = sum_n dx (B u + A a x_prev) = u * sum_n dx B + sum_n A dx a x_prev;  through the softplus: * sigmoid
  o.ddraw = fmaf(uu, o.du_acc, dd_acc) * sg;
  return o;
}

// Sum of the 8 values of `adjoint_step` over the wave's 16 channels: three reduce-scatter levels on packed pairs (one
// v_pk_add_f32 per two sums; lane bits 5 and 4 by the cross-row swaps), the last two inside a 16-lane row as DPP adds
// whose bank masks pick the half that keeps each value -- no select.  EVERY lane ends with a total: value (lane >> 3) & 7 of
// quad q, the same in lanes i and i ^ 4, so the caller's store needs no lane predicate (a same-address LDS write costs
// what a plain one does: tools/probe/valu_cost.hip) and the step stays one basic block.
__device__ __forceinline__ float chan_sum8(const sf2 (&v)[4]) {
  sf2 r[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {                    // lane bit 5: dB pair <-> dC pair
    auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[h].x), __float_as_uint(v[2 + h].x), false, false);
    auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[h].y), __float_as_uint(v[2 + h].y), false, false);
    const sf2 lo = {__uint_as_float(sx[0]), __uint_as_float(sy[0])}, hi = {__uint_as_float(sx[1]), __uint_as_float(sy[1])};
    r[h] = lo + hi;
  }
  auto tx = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0].x), __float_as_uint(r[1].x), false, false);   // lane bit 4
  auto ty = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[0].y), __float_as_uint(r[1].y), false, false);
  const sf2 lo = {__uint_as_float(tx[0]), __uint_as_float(ty[0])}, hi = {__uint_as_float(tx[1]), __uint_as_float(ty[1])};
  const sf2 t = lo + hi;
  // lane bit 3 (row_ror:8 == lane ^ 8): lanes with the bit clear (banks 0, 1) keep t.x, the others t.y;
  // lane bit 2: lanes with the bit clear (banks 0, 2) add lane i + 4 (row_ror:12), the others lane i - 4 (row_ror:4)
  float u, w;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %1, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(u), "=&v"(w) : "v"(t.x), "v"(t.y));
  return w;
}

