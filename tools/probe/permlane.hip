// probe: semantics of v_permlane32_swap / v_permlane16_swap and DPP row_ror on gfx950 (scan reduce-scatter)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* o) {
  float a = threadIdx.x, b = 100.f + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  o[threadIdx.x] = __uint_as_float(r[0]);
  o[64 + threadIdx.x] = __uint_as_float(r[1]);
  o[128 + threadIdx.x] = __uint_as_float(r2[0]);
  o[192 + threadIdx.x] = __uint_as_float(r2[1]);
  o[256 + threadIdx.x] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x124, 0xf, 0xf, true));  // row_ror:4
  o[320 + threadIdx.x] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x12C, 0xf, 0xf, true));  // row_ror:12
}
int main() {
  float* d; (void)hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[384]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1", "row_ror:4", "row_ror:12"};
  for (int t = 0; t < 6; ++t) { printf("%s:", names[t]); for (int i = 0; i < (t < 4 ? 64 : 16); i += (t < 4 ? 8 : 1)) printf(" %g", h[t * 64 + i]); printf("\n"); }
  return 0;
}
