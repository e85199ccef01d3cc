// Probe: how many loads can ONE wave keep in flight?  Compares global_load_lds_dwordx4 (LDS-DMA) with plain
// global_load_dwordx4 into registers: each wave streams its own contiguous slice of an HBM-cold buffer, U loads of
// 1 KiB per wave issued back to back, then s_waitcnt vmcnt(0).  Usage: lds_dma   (prints a table)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, float* out, long per_wave, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long wave_id = (long)blockIdx.x * 4 + wv;
  const char* p = src + wave_id * per_wave + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        typedef __attribute__((address_space(1))) const void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        __builtin_amdgcn_global_load_lds((gptr)(p + (long)(it * U + u) * 1024), (lptr)(smem + (wv * U + u) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(p + (long)(it * U + u) * 1024);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if (MODE == 0) acc = *reinterpret_cast<u32x4*>(smem + tid * 16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345677u) out[0] = 1.f;
}

// GEMM-like pattern: a wave owns panels of 32 rows x rowbytes and walks them K chunk by K chunk: every 1 KiB wave load
// covers 1024/seg rows x seg contiguous bytes (seg = 128: one 64-wide bf16 K step of 8 rows).
template <int U>
__global__ __launch_bounds__(256) void strided_kernel(const char* __restrict__ src, float* out, int panels, int rowbytes, int seg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long wave_id = (long)blockIdx.x * 4 + wv;
  const int rpl = 1024 / seg;                       // rows per wave load
  const int lrow = (lane * 16) / seg, lcol = (lane * 16) % seg;
  int n = 0;
  for (int pn = 0; pn < panels; ++pn) {
    const char* base = src + ((wave_id * panels + pn) * 32) * (long)rowbytes;
    for (int c = 0; c < rowbytes / seg; ++c)
      for (int g = 0; g < 32 / rpl; ++g) {
        typedef __attribute__((address_space(1))) const void* gptr;
        typedef __attribute__((address_space(3))) void* lptr;
        __builtin_amdgcn_global_load_lds((gptr)(base + (long)(g * rpl + lrow) * rowbytes + c * seg + lcol),
                                         (lptr)(smem + (wv * U + (n % U)) * 1024), 16, 0, 0);
        if (++n % U == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  u32x4 acc = *reinterpret_cast<u32x4*>(smem + tid * 16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345677u) out[0] = 1.f;
}

template <int U>
void run_strided(const char* buf, float* out, int wg_per_cu, size_t total, int rowbytes, int seg) {
  const int G = 256 * wg_per_cu;
  const int panels = (int)(total / ((size_t)G * 4 * 32 * rowbytes));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL((strided_kernel<U>), dim3(G), dim3(256), 4 * U * 1024 > 4096 ? 4 * U * 1024 : 4096, 0, buf, out, panels, rowbytes, seg);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (r && ms < best) best = ms;
  }
  const double bytes = (double)panels * 32 * rowbytes * G * 4;
  printf("strided rowbytes=%d seg=%d U=%d wg/cu=%d  %7.1f us  %7.0f GB/s\n", rowbytes, seg, U, wg_per_cu, best * 1e3, bytes / best / 1e6);
}

template <int U, int MODE>
void run(const char* buf, float* out, int wg_per_cu, size_t total) {
  const int G = 256 * wg_per_cu;
  const long per_wave = (long)(total / ((size_t)G * 4)) / (U * 1024) * (U * 1024);
  const int iters = (int)(per_wave / (U * 1024));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL((stream_kernel<U, MODE>), dim3(G), dim3(256), 4 * U * 1024 > 4096 ? 4 * U * 1024 : 4096, 0, buf, out, per_wave, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (r && ms < best) best = ms;
  }
  const double bytes = (double)per_wave * G * 4;
  printf("%s U=%d wg/cu=%d  %7.1f us  %7.0f GB/s  (%.1f GB/s per wave)\n", MODE == 0 ? "lds-dma" : "regs   ", U, wg_per_cu, best * 1e3,
         bytes / best / 1e6, bytes / best / 1e6 / (G * 4));
}

int main() {
  const size_t total = (size_t)768 << 20;      // larger than the 256 MB memory-side cache
  char* buf; float* out;
  hipMalloc(&buf, total); hipMalloc(&out, 4);
  hipMemset(buf, 1, total);
  for (int wg = 1; wg <= 1; wg *= 2) {
    run<1, 0>(buf, out, wg, total); run<2, 0>(buf, out, wg, total); run<4, 0>(buf, out, wg, total); run<8, 0>(buf, out, wg, total);
    run<1, 1>(buf, out, wg, total); run<2, 1>(buf, out, wg, total); run<4, 1>(buf, out, wg, total); run<8, 1>(buf, out, wg, total);
  }
  for (int wg = 1; wg <= 2; ++wg)
    for (int rb : {1536, 768, 384})
      for (int seg : {128, 256, 384, 512, 768, 1536}) {
        if (seg > rb || rb % seg || 1024 % seg) { if (!(seg == rb && 1024 % seg)) continue; else continue; }
        run_strided<4>(buf, out, wg, total, rb, seg);
        run_strided<8>(buf, out, wg, total, rb, seg);
      }
  return 0;
}
