// How much HBM read bandwidth a K-slow GEMM operand pattern gets: a (rows x row_bytes) matrix is read by workgroups that
// each own a W-byte column chunk of a contiguous range of rows (the 128-column tile of a weight-gradient GEMM is W = 256).
// hipcc --offload-arch=gfx950 -O3 tools/probe/read_chunks.hip -o /tmp/read_chunks && /tmp/read_chunks
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ p, int row_bytes, int W, int rows_per_blk, int chunks, unsigned* out) {
  const int chunk = blockIdx.x % chunks, slab = blockIdx.x / chunks;
  const int lanes_per_row = W / 16, rows_per_pass = 256 / lanes_per_row;
  const int lr = threadIdx.x % lanes_per_row, r0 = threadIdx.x / lanes_per_row;
  const char* base = p + (size_t)slab * rows_per_blk * row_bytes + (size_t)chunk * W + lr * 16;
  unsigned acc = 0;
  for (int r = r0; r + 3 * rows_per_pass < rows_per_blk; r += 4 * rows_per_pass) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (size_t)(r + u * rows_per_pass) * row_bytes);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const int row_bytes = 1536;
  const size_t rows = 25088 * 48;                       // 1.85 GB
  const size_t bytes = rows * row_bytes;
  char* p; unsigned* o;
  (void)hipMalloc(&p, bytes); (void)hipMalloc(&o, 4); (void)hipMemset(p, 1, bytes);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int W : {128, 256, 384, 512, 768, 1536})
    for (int rpb : {512, 3584}) {
      const int chunks = row_bytes / W, blocks = (int)(rows / rpb) * chunks;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, p, row_bytes, W, rpb, chunks, o);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (rep) printf("chunk %4d B, %4d rows per block, %6d blocks: %.2f TB/s\n", W, rpb, blocks, 3.0 * bytes / ms / 1e9);
      }
    }
  return 0;
}
