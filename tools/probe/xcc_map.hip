// Is the workgroup -> XCD placement stable across launches?  Every workgroup records HW_REG_XCC_ID; the program prints,
// per launch, which physical XCD each class (blockIdx % 8) landed on, for 1-D and 2-D grids of several sizes, with other
// kernels (different grid sizes) launched in between, eagerly and replayed from a HIP graph.
// Then: does a consumer kernel that reads what a producer wrote run faster when the same class handles the same data?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probe/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(int* out) {
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0) out[lin] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;      // HW_REG_XCC_ID = 20, bits [3:0]
}
__global__ void filler(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
// producer: workgroup w writes chunk c = map(w); consumer: workgroup w reads chunk c = map(w ^ shift) (shift != 0: another class)
__global__ __launch_bounds__(256) void prod(uint4* buf, int chunk_vecs) {
  uint4* p = buf + (size_t)blockIdx.x * chunk_vecs;
  for (int i = threadIdx.x; i < chunk_vecs; i += 256) p[i] = make_uint4(i, blockIdx.x, 3, 4);
}
__global__ __launch_bounds__(256) void cons(const uint4* buf, int chunk_vecs, int shift, unsigned* out) {
  const int c = (blockIdx.x + shift) % gridDim.x;
  const uint4* p = buf + (size_t)c * chunk_vecs;
  unsigned acc = 0;
  for (int i = threadIdx.x; i < chunk_vecs; i += 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) out[0] = acc;
}
static void show(const char* tag, const std::vector<int>& h, int n) {
  int cls[8][8] = {};
  for (int i = 0; i < n; ++i) cls[i % 8][h[i] & 7]++;
  printf("%-34s class->xcc:", tag);
  for (int c = 0; c < 8; ++c) {
    int best = 0;
    for (int x = 1; x < 8; ++x) if (cls[c][x] > cls[c][best]) best = x;
    printf(" %d(%3.0f%%)", best, 100.0 * cls[c][best] / ((n + 7 - c) / 8));
  }
  printf("\n");
}
int main() {
  int* d; float* f; hipMalloc(&d, 1 << 20); hipMalloc(&f, 64 << 20);
  std::vector<int> h(1 << 18);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 3; ++rep)
    for (int n : {256, 392, 1792, 512}) {
      hipLaunchKernelGGL(filler, dim3(1000 + 37 * rep), dim3(256), 0, st, f, 1 << 20);
      hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, st, d);
      hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
      char tag[64]; snprintf(tag, 64, "eager rep %d grid %d", rep, n); show(tag, h, n);
    }
  {   // 2-D grid (14, 128) as conv_pool_fwd launches
    hipLaunchKernelGGL(who, dim3(14, 128), dim3(64), 0, st, d);
    hipMemcpyAsync(h.data(), d, 1792 * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
    show("eager 2-D grid (14, 128)", h, 1792);
  }
  {   // graph: filler, who(392) -> d, filler, who(1792) -> d + 4096, replayed 3 times
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    hipLaunchKernelGGL(filler, dim3(777), dim3(256), 0, st, f, 1 << 20);
    hipLaunchKernelGGL(who, dim3(392), dim3(64), 0, st, d);
    hipLaunchKernelGGL(filler, dim3(1234), dim3(256), 0, st, f, 1 << 20);
    hipLaunchKernelGGL(who, dim3(1792), dim3(64), 0, st, d + 4096);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 3; ++rep) {
      hipGraphLaunch(ge, st);
      hipMemcpyAsync(h.data(), d, (4096 + 1792) * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
      char tag[64]; snprintf(tag, 64, "graph replay %d grid 392", rep); show(tag, h, 392);
      std::vector<int> h2(h.begin() + 4096, h.begin() + 4096 + 1792);
      snprintf(tag, 64, "graph replay %d grid 1792", rep); show(tag, h2, 1792);
    }
  }
  // producer -> consumer with the same / another class reading a chunk: 256 workgroups x 75 KB (one half image of U)
  uint4* buf; unsigned* o; hipMalloc(&buf, (size_t)512 * 150528); hipMalloc(&o, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int wgs : {256, 512})
    for (int shift : {0, 1, 3, 8, 0, 1}) {
      const int vecs = 75264 / 16;
      float tot = 0;
      for (int rep = 0; rep < 20; ++rep) {
        hipLaunchKernelGGL(filler, dim3(4096), dim3(256), 0, st, f, 16 << 20);      // 64 MB of other traffic in between
        hipLaunchKernelGGL(prod, dim3(wgs), dim3(256), 0, st, buf, vecs);
        hipEventRecord(a, st);
        hipLaunchKernelGGL(cons, dim3(wgs), dim3(256), 0, st, buf, vecs, shift, o);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (rep >= 4) tot += ms;
      }
      printf("consumer after producer, %d workgroups x 75 KB, chunk of workgroup + %d: %.2f us\n", wgs, shift, tot / 16 * 1e3);
    }
  return 0;
}
