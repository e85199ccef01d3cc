// Read-only HBM stream ceiling: every thread sums 16-byte vectors of a buffer far larger than the Infinity Cache.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probe/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int U>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ p, size_t n, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)1850 << 20;
  uint4* p; unsigned* o;
  hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {512, 1024, 2048, 4096, 8192}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(rd<8>, dim3(blocks), dim3(256), 0, 0, p, bytes / 16, o);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep) printf("U=8 blocks %5d: %.2f TB/s\n", blocks, 5.0 * bytes / ms / 1e9);
    }
  }
  for (int blocks : {1024, 4096}) {
    hipEventRecord(a);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(rd<2>, dim3(blocks), dim3(256), 0, 0, p, bytes / 16, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("U=2 blocks %5d: %.2f TB/s\n", blocks, 5.0 * bytes / ms / 1e9);
  }
  return 0;
}
