// Calibration probe for the WRITE_SIZE counter (round 5, verdict item 4d): three kernels store EXACTLY the same number of
// bytes -- 64 MiB -- with 4-byte, 8-byte and 16-byte stores per lane (coalesced: a wave writes 256 / 512 / 1024 contiguous
// bytes), and one stores 4 bytes per lane at a 12-byte lane stride (the row walkers' bf16 triple).  rocprofv3 --pmc
// WRITE_SIZE on this binary says what the counter reports for each width; tools/probe/write_size.sh prints the ratios.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/write_size.hip -o /tmp/write_size
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void store_b32(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void store_b64(uint2* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint2((unsigned)i, 1u);
}
__global__ void store_b128(uint4* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
// three dwords per lane at a 12-byte lane stride, as three separate 4-byte stores (every byte written exactly once)
__global__ void store_b32x3_stride12(unsigned* p, size_t n3) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x) {
    p[3 * i] = (unsigned)i; p[3 * i + 1] = 1u; p[3 * i + 2] = 2u;
  }
}

int main() {
  const size_t bytes = 64ull << 20;
  void* d;
  if (hipMalloc(&d, bytes + 64) != hipSuccess) return 1;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(store_b32, dim3(4096), dim3(256), 0, 0, (unsigned*)d, bytes / 4);
    hipLaunchKernelGGL(store_b64, dim3(4096), dim3(256), 0, 0, (uint2*)d, bytes / 8);
    hipLaunchKernelGGL(store_b128, dim3(4096), dim3(256), 0, 0, (uint4*)d, bytes / 16);
    hipLaunchKernelGGL(store_b32x3_stride12, dim3(4096), dim3(256), 0, 0, (unsigned*)d, bytes / 12);
  }
  hipDeviceSynchronize();
  printf("stored %zu bytes per kernel\n", bytes);
  return 0;
}
