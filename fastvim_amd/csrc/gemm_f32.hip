// fp32-MFMA GEMM (v_mfma_f32_16x16x4_f32): the projections in the reference's DEFAULT precision (fp32,
// imagenet_classification/train.py:17) and every shape the tuned bf16 kernels of gemm_mfma.hip do not take (odd
// extents, unaligned rows, token counts that are not multiples of 64).  Replaces the rocBLAS / hipBLASLt calls behind
// F.linear / matmul / bmm in mamba_simple_faster.py:189-193, 321-327, 435-444 and models/fastvim.py:95, 537 when the model
// runs in fp32, so that the fp32 golden tests pin this build's own tiling and epilogue code.
//
//   C_b[M][N] = sum_k A_b(m,k) * B_b(k,n) (+ bias[n]),   b = 0 .. batch-1 (grid.z), fp32 accumulate
//   a_k_slow = 0: A(m,k) = A[m*lda + k]      a_k_slow = 1: A(m,k) = A[k*lda + m]
//   b_k_slow = 0: B(k,n) = B[n*ldb + k]      b_k_slow = 1: B(k,n) = B[k*ldb + n]
// Operands fp32 or bf16 (bf16 is widened exactly on the way into LDS: the products and sums are the fp32 MFMA's either
// way), C fp32 or bf16.  A deterministic split-K weight gradient is the batched form: batch = slices, strides = one K
// slice, C = the fp32 partials (summed in fixed order by fv_reduce_partials).
//
// gfx950 structure: 256 threads = 2 x 2 wave64; a wave owns a (BM/2) x (BN/2) accumulator of 16 x 16 MFMA tiles; K tiles
// of 16 go global -> registers -> LDS (the loads of tile t+1 are in flight while tile t is multiplied) and are kept
// [k][extent + 16] in LDS -- a fragment read (lane -> element lane % 16 of k row lane / 16) is conflict-free for both
// operands whatever their layout in HBM.  The MFMA is issued with the roles swapped (B fragment in the A slot), which
// leaves a lane with 4 consecutive output columns of one row: 16-byte C stores.  One instruction is 16 x 16 x 4 x 2 flops
// in 32 cycles; 4 waves per CU give the 157 TFLOP/s fp32 matrix peak when the loads stay hidden.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BKF = 16;

struct GemmF32Params {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC;       // batch strides (elements)
  int c_bf16;
  int vecA, vecB;        // rows start 16-byte (fp32) / 8-byte (bf16) aligned and the contiguous extent is a multiple of 4
};

template <typename T>
__device__ __forceinline__ float4 ld4(const T* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

// One operand's K tile: EXT x 16 elements as EXT * 4 quads of 4 consecutive elements along the contiguous axis.
//   K-contiguous (KS = 0): quad q -> row q / 4, k = (q % 4) * 4      K-slow (KS = 1): quad q -> k = q / (EXT / 4), col (q % (EXT / 4)) * 4
template <typename T, int KS, int EXT>
struct TileF32 {
  static constexpr int NQ = EXT * BKF / 4 / 256;
  float4 v[NQ];
  __device__ __forceinline__ void load(const T* base, long ld, int r0, int k0, int rmax, int kmax, bool vec, int tid) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * 256;
      int r, k;
      if (KS == 0) { r = q >> 2; k = (q & 3) * 4; }
      else { k = q / (EXT / 4); r = (q % (EXT / 4)) * 4; }
      const int gr = r0 + r, gk = k0 + k;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KS == 0) {
        if (gr < rmax) {
          const T* p = base + (long)gr * ld + gk;
          if (vec && gk + 4 <= kmax) t = ld4<T>(p);
          else {
            float* tt = &t.x;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gk + j < kmax) tt[j] = io<T>::ld(p + j);
          }
        }
      } else {
        if (gk < kmax) {
          const T* p = base + (long)gk * ld + gr;
          if (vec && gr + 4 <= rmax) t = ld4<T>(p);
          else {
            float* tt = &t.x;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gr + j < rmax) tt[j] = io<T>::ld(p + j);
          }
        }
      }
      v[i] = t;
    }
  }
  // LDS image: [k][EXT + 16] fp32
  __device__ __forceinline__ void store(float* lds, int tid) const {
    constexpr int RS = EXT + 16;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = tid + i * 256;
      if (KS == 0) {
        const int r = q >> 2, k = (q & 3) * 4;
        lds[(k + 0) * RS + r] = v[i].x;
        lds[(k + 1) * RS + r] = v[i].y;
        lds[(k + 2) * RS + r] = v[i].z;
        lds[(k + 3) * RS + r] = v[i].w;
      } else {
        const int k = q / (EXT / 4), r = (q % (EXT / 4)) * 4;
        *reinterpret_cast<float4*>(lds + k * RS + r) = v[i];
      }
    }
  }
};

template <typename TA, typename TB, int AKS, int BKS, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmF32Params p) {
  constexpr int RSA = BM + 16, RSB = BN + 16;
  constexpr int MT = BM / 32, NT_ = BN / 32;      // 16 x 16 tiles per wave along m / n
  __shared__ __attribute__((aligned(16))) float sA[2][BKF * RSA];
  __shared__ __attribute__((aligned(16))) float sB[2][BKF * RSB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const TA* A = reinterpret_cast<const TA*>(p.A) + (long)blockIdx.z * p.sA;
  const TB* B = reinterpret_cast<const TB*>(p.B) + (long)blockIdx.z * p.sB;
  const bool vA = p.vecA != 0, vB = p.vecB != 0;

  f32x4 acc[NT_][MT];      // [n tile][m tile]; acc[..][v] = C[m = .. + lane % 16][n = .. + 4 * (lane / 16) + v]
#pragma unroll
  for (int a = 0; a < NT_; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  TileF32<TA, AKS, BM> ra;
  TileF32<TB, BKS, BN> rb;
  const int nt = (p.K + BKF - 1) / BKF;
  ra.load(A, p.lda, m0, 0, p.M, p.K, vA, tid);
  rb.load(B, p.ldb, n0, 0, p.N, p.K, vB, tid);
  ra.store(sA[0], tid);
  rb.store(sB[0], tid);
  __syncthreads();
  const int fo = (lane >> 4) * 1, fx = lane & 15;      // fragment: k row lane / 16 of a 4-deep step, element lane % 16
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      ra.load(A, p.lda, m0, (t + 1) * BKF, p.M, p.K, vA, tid);
      rb.load(B, p.ldb, n0, (t + 1) * BKF, p.N, p.K, vB, tid);
    }
    const float* a_ = sA[cur] + wm * (BM / 2) + fx;
    const float* b_ = sB[cur] + wn * (BN / 2) + fx;
#pragma unroll
    for (int ks = 0; ks < BKF / 4; ++ks) {
      float fa[MT], fb[NT_];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[i] = a_[(ks * 4 + fo) * RSA + i * 16];
#pragma unroll
      for (int i = 0; i < NT_; ++i) fb[i] = b_[(ks * 4 + fo) * RSB + i * 16];
#pragma unroll
      for (int a = 0; a < NT_; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[a], fa[b], acc[a][b], 0, 0, 0);
    }
    if (t + 1 < nt) {
      ra.store(sA[cur ^ 1], tid);
      rb.store(sB[cur ^ 1], tid);
    }
    __syncthreads();
  }
  // epilogue
  char* Cb = reinterpret_cast<char*>(p.C);
  const long cz = (long)blockIdx.z * p.sC;
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m0 + wm * (BM / 2) + b * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < NT_; ++a) {
      const int n = n0 + wn * (BN / 2) + a * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[a][b];
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < p.N) v[j] += p.bias[n + j];
      }
      const long off = cz + (long)m * p.ldc + n;
      if (p.c_bf16) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(Cb) + off;
        if (n + 3 < p.N && ((reinterpret_cast<uintptr_t>(dst) & 7) == 0)) {
          uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<uint2*>(dst) = pk;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < p.N) dst[j] = __float2bfloat16(v[j]);
        }
      } else {
        float* dst = reinterpret_cast<float*>(Cb) + off;
        if (n + 3 < p.N && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) *reinterpret_cast<f32x4*>(dst) = v;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < p.N) dst[j] = v[j];
        }
      }
    }
  }
}

template <typename TA, typename TB, int AKS, int BKS>
void launch_f32(const GemmF32Params& p, int batch, hipStream_t st) {
  // 128 x 128 tiles where they fill the chip twice over, 64 x 64 below (projections of a few thousand rows, x_proj, the head)
  const long big = (long)fv_cdiv(p.M, 128) * fv_cdiv(p.N, 128) * batch;
  if (big >= 512 && p.N >= 96) {
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, AKS, BKS, 128, 128>), dim3(fv_cdiv(p.M, 128) * fv_cdiv(p.N, 128), 1, batch), dim3(256),
                       0, st, p);
  } else {
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, AKS, BKS, 64, 64>), dim3(fv_cdiv(p.M, 64) * fv_cdiv(p.N, 64), 1, batch), dim3(256), 0,
                       st, p);
  }
}

template <typename TA, typename TB>
void dispatch_layout(const GemmF32Params& p, int a_ks, int b_ks, int batch, hipStream_t st) {
  if (!a_ks && !b_ks) launch_f32<TA, TB, 0, 0>(p, batch, st);
  else if (!a_ks && b_ks) launch_f32<TA, TB, 0, 1>(p, batch, st);
  else if (a_ks && b_ks) launch_f32<TA, TB, 1, 1>(p, batch, st);
  else launch_f32<TA, TB, 1, 0>(p, batch, st);
}

bool vec_ok(const void* base, long ld, long stride, int batch, int dtype) {
  const uintptr_t al = dtype == FV_BF16 ? 8 : 16, es = dtype == FV_BF16 ? 2 : 4;
  if (reinterpret_cast<uintptr_t>(base) % al) return false;
  if ((ld * es) % al) return false;
  if (batch > 1 && (stride * es) % al) return false;
  return true;
}

}  // namespace

extern "C" int fv_gemm_f32(const void* A, int a_dtype, const void* B, int b_dtype, void* C, int c_dtype, const float* bias,
                           int M, int N, int K, long lda, long ldb, long ldc, int a_k_slow, int b_k_slow, int batch,
                           long strideA, long strideB, long strideC, fv_stream_t stream) {
  FV_CHECK(A && B && C, "gemm_f32: null pointer");
  FV_CHECK(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "gemm_f32: bad extents");
  FV_CHECK((a_dtype == FV_F32 || a_dtype == FV_BF16) && (b_dtype == FV_F32 || b_dtype == FV_BF16) &&
               (c_dtype == FV_F32 || c_dtype == FV_BF16),
           "gemm_f32: operands fp32 or bf16, C fp32 or bf16");
  FV_CHECK(batch <= 65535, "gemm_f32: batch <= 65535");
  if (M == 0 || N == 0) return FV_OK;
  GemmF32Params p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.sA = strideA; p.sB = strideB; p.sC = strideC;
  p.c_bf16 = c_dtype == FV_BF16;
  p.vecA = vec_ok(A, lda, strideA, batch, a_dtype);
  p.vecB = vec_ok(B, ldb, strideB, batch, b_dtype);
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == FV_F32 && b_dtype == FV_F32) dispatch_layout<float, float>(p, a_k_slow, b_k_slow, batch, st);
  else if (a_dtype == FV_BF16 && b_dtype == FV_BF16) dispatch_layout<bf16_t, bf16_t>(p, a_k_slow, b_k_slow, batch, st);
  else if (a_dtype == FV_F32) dispatch_layout<float, bf16_t>(p, a_k_slow, b_k_slow, batch, st);
  else dispatch_layout<bf16_t, float>(p, a_k_slow, b_k_slow, batch, st);
  FV_LAUNCH_CHECK();
  return FV_OK;
}
