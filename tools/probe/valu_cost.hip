// probe: issue cost of the scan backward's instruction mix on gfx950 with W waves per SIMD -- cycles (s_memtime) per
// wave-instruction on one SIMD = elapsed / (waves per SIMD x instructions per wave); 8 independent chains per wave.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define REP 8
#define ITERS 2000
#define STR2(x) #x
#define STR(x) STR2(x)
template <int OP>
__global__ __launch_bounds__(1024) void k(float* o, unsigned long long* t, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[REP], b = {seed, seed * 0.5f};
  for (int i = 0; i < REP; ++i) a[i] = (f2){seed + i + threadIdx.x, seed - i};
  __shared__ float4 lds[1024];
  lds[threadIdx.x] = make_float4(seed, 1.f, 2.f, 3.f);
  __syncthreads();
  const unsigned lad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4*)lds + (threadIdx.x & 63) * 16;
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float4*)lds;
  const unsigned ladw = lbase + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63) * 4;          // a wave's own 256 bytes, one dword per lane
  const unsigned lad4 = lbase + (threadIdx.x >> 6) * 256 + ((threadIdx.x & 63) >> 2) * 4;   // four lanes per dword (same value)
  const unsigned ladq = lbase + (threadIdx.x >> 6) * 256 + ((threadIdx.x & 63) >> 2) * 16;  // four lanes per dword, stride 16 B
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
      if (OP == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y));
      if (OP == 5) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[i].y));
      if (OP == 6) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i].x));
      if (OP == 7) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i].x));
      if (OP == 8) { float4 r; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lad) : "memory"); a[i].x += r.x; }
      if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(b.x));
      if (OP == 11) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b)); asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 4) % REP].y)); }
      if (OP == 12) asm volatile("ds_write_b32 %0, %1" :: "v"(lad), "v"(a[i].x) : "memory");
      if (OP == 14) asm volatile("ds_write_b32 %0, %1" :: "v"(lad4), "v"(a[i].x) : "memory");
      if (OP == 15) asm volatile("ds_write_b32 %0, %1" :: "v"(ladw), "v"(a[i].x) : "memory");
      if (OP == 16) asm volatile("ds_write_b32 %0, %1" :: "v"(ladq), "v"(a[i].x) : "memory");
      if (OP == 13) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(b));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < REP; ++i) s += a[i].x + a[i].y;
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, int waves, float* o, unsigned long long* t) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, o, t, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, o, t, 1.0f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += h[i];
  m /= 256;
  const int per = (OP == 11 ? 2 : 1) * REP * ITERS;
  printf("%-34s %2d waves/CU: %8.0f ticks (%7.1f us by events: %5.2f ticks/ns), %6.2f ticks = %5.2f ns per wave-instruction on its SIMD\n", name, waves, m,
         ms * 1e3, m / (ms * 1e6), m / (per * (waves / 4.0)), ms * 1e6 / (per * (waves / 4.0)));
}
int main() {
  float* o; unsigned long long* t;
  (void)hipMalloc(&o, 256 * 1024 * 4); (void)hipMalloc(&t, 256 * 8);
  for (int waves : {4, 12}) {
    run<0>("v_fma_f32", waves, o, t);
    run<1>("v_pk_fma_f32", waves, o, t);
    run<2>("v_pk_mul_f32", waves, o, t);
    run<13>("v_pk_mul_f32 op_sel_hi", waves, o, t);
    run<9>("v_pk_add_f32", waves, o, t);
    run<10>("v_mov_b32", waves, o, t);
    run<3>("v_exp_f32", waves, o, t);
    run<11>("v_pk_fma_f32 + v_exp_f32 pair", waves, o, t);
    run<4>("v_permlane32_swap_b32", waves, o, t);
    run<5>("v_permlane16_swap_b32", waves, o, t);
    run<6>("v_add_f32_dpp quad_perm", waves, o, t);
    run<7>("v_add_f32_dpp row_ror:8", waves, o, t);
    run<8>("ds_read_b128 + wait", waves, o, t);
    run<12>("ds_write_b32", waves, o, t);
    run<15>("ds_write_b32 own 256 B per wave", waves, o, t);
    run<14>("ds_write_b32 4 lanes per dword", waves, o, t);
    run<16>("ds_write_b32 4 lanes/dword, 16 B apart", waves, o, t);
  }
  return 0;
}
