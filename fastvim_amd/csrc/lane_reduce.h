// Cross-lane helpers shared by the pooled scan (scan_cl.hip) and the short-sequence op scan (scan_bdl.hip):
// lane = (channel, state quad) -- the 4 lanes of a channel are adjacent, a wave covers 16 channels.
#pragma once
#include "common.h"

__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));  // [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));  // [2,3,0,1]
  return v;
}

// Reduce-scatter over the 16 channel-lanes of a wave (lane bits 2..5); PV values per lane, PV in {16, 32}.
// On return the lane whose channel index is c holds the totals of value indices [c*PV/16, (c+1)*PV/16).
// Lane bits 5 and 4 use the gfx950 cross-row swaps (v_permlane32_swap / v_permlane16_swap: one swap + one
// add per value pair, no select); bits 3 and 2 stay inside a 16-lane row: DPP row rotates.
template <int CTRL>
__device__ __forceinline__ float add_dpp(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int H, int PV>
__device__ __forceinline__ void rs_swap32(float (&v)[PV]) {
#pragma unroll
  for (int e = 0; e < H; ++e) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[e]), __float_as_uint(v[e + H]), false, false);
    v[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lanes < 32: v[e] + partner's; lanes >= 32: v[e+H] pair
  }
}
template <int H, int PV>
__device__ __forceinline__ void rs_swap16(float (&v)[PV]) {
#pragma unroll
  for (int e = 0; e < H; ++e) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[e]), __float_as_uint(v[e + H]), false, false);
    v[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
template <int H, int PV>
__device__ __forceinline__ void rs_row8(float (&v)[PV], int lane) {
  const bool up = lane & 8;
#pragma unroll
  for (int e = 0; e < H; ++e) {
    const float lo = add_dpp<0x128>(v[e]), hi = add_dpp<0x128>(v[e + H]);      // row_ror:8 == lane ^ 8
    v[e] = up ? hi : lo;
  }
}
template <int H, int PV>
__device__ __forceinline__ void rs_row4(float (&v)[PV], int lane) {
  const bool up = lane & 4;
#pragma unroll
  for (int e = 0; e < H; ++e) {
    const float lo = add_dpp<0x12C>(v[e]);          // row_ror:12: lane i reads lane i+4 (valid where bit 2 is clear)
    const float hi = add_dpp<0x124>(v[e + H]);      // row_ror:4 : lane i reads lane i-4 (valid where bit 2 is set)
    v[e] = up ? hi : lo;
  }
}
template <int PV>
__device__ __forceinline__ void chan_reduce_scatter(float (&v)[PV], int lane) {
  rs_swap32<PV / 2, PV>(v);
  rs_swap16<PV / 4, PV>(v);
  rs_row8<PV / 8, PV>(v, lane);
  rs_row4<PV / 16, PV>(v, lane);
}


// Value `v` of lane i plus that of lane i ^ 4 (the last level of a channel reduce whose values are already scattered: both
// lanes end with the total).  Two DPP adds whose bank masks pick the direction: lanes with bit 2 clear (banks 0, 2) read
// lane i + 4 (row_ror:12), the others lane i - 4 (row_ror:4).
__device__ __forceinline__ float add_lane_xor4(float v) {
  float w;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(w) : "v"(v));
  return w;
}
// lane k of every quad's value, in all four lanes of the quad (quad_perm [k, k, k, k])
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xf, 0xf, true));
}
// Sum of 8 values per lane over the 16 channel lanes of a wave: three reduce-scatter levels (lane bits 5, 4, 3) and one
// plain level (bit 2).  Every lane returns the total of value index (lane >> 3) & 7, the same in lanes i and i ^ 4.
__device__ __forceinline__ float chan_sum8_plain(float (&v)[8], int lane) {
  rs_swap32<4, 8>(v);
  rs_swap16<2, 8>(v);
  rs_row8<1, 8>(v, lane);
  return add_lane_xor4(v[0]);
}
