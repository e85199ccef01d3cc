// Pieces shared by the forward and backward fused-mixer kernels.
#pragma once
#include "rowwalk.h"

namespace fvi {   // shared between the mixer translation units

struct BwdParams {
  const void *xz, *dg, *skip, *dob_in;
  const float *wf, *bf, *wb, *bb, *Df, *Db, *lnw, *lnb, *mean, *rstd, *dxc, *yc;
  const void* amax;                     // max pooling: argmax columns saved by the forward (else null)
  const void* dxc2;                     // conv_pool_bwd, nullable: second addend of the pooled gradient, storage dtype (the
                                        // other channel chunk's x_proj term of fv_mixer_scan_bwd_xproj); whole-row kernel only
  void *dxz, *dob;
  float *dyc, *part;
  Geo geo;
  int B, d_in, use_norm;
  float pool_scale;
};

struct FwdParams {
  const void* xz;                       // (B, L, 2*d_in)
  const float *wf, *bf, *wb, *bb;       // conv1d / conv1d_b: (d_in, CW), (d_in)
  void* xc;                             // (2, B, rows*tpp, d_in) pooled conv output [dir 0 = fwd]
  void* skip;                           // (B, L, d_in) D*conv_f + D_b*conv_b, memory token order (nullable in conv_pool)
  void* amax;                           // (2, B, rows*tpp, d_in) max pooling: column of the maximum (storage dtype; nullable)
  const float* yc;                      // (2, B, rows*tpp, d_in) scan output
  const float *Df, *Db, *lnw, *lnb;     // (d_in)
  void* g;                              // (B, L, d_in) gated LayerNorm output
  float *mean, *rstd;                   // (B*L) LayerNorm statistics (saved for backward)
  Geo geo;
  int B, d_in;
  float pool_scale;                     // scaling_factor / cols (mean) or 1 (max)
  float eps;
  int use_norm;
};

// Whole-row conv+pool(+skip) forward (convpool_fwd_row.hip); FV_ERR_UNSUPPORTED -> caller runs the generic kernel.
int conv_pool_fwd_row(const FwdParams& p, int pool_max, int dtype, hipStream_t st);

constexpr int RGMAX = 4;   // a block walks up to RGMAX pooling rows concurrently (one per row group) and emits ONE partial

// Whole-row conv+pool backward (convpool_bwd_row.hip).  Same grid / partial layout as the generic kernel;
// returns FV_ERR_UNSUPPORTED when the shape is not one it is built for (the caller then runs the generic one).
int conv_pool_bwd_row(const BwdParams& p, int nch, int rg, int grid, size_t smem, int dtype, hipStream_t st);

// Wave-per-token combine kernels (combine_wave.hip); FV_ERR_UNSUPPORTED -> caller runs the generic kernel.
// combine_wave_blocks: grid (= rows of dLN partials) of the backward, 0 when the wave kernels do not apply.
int combine_wave_blocks(int B, int rows, int tpp, int d_in);
int combine_fwd_wave(const FwdParams& p, int dtype, hipStream_t st);
int combine_bwd_wave(const BwdParams& p, int dtype, hipStream_t st);

}  // namespace fvi

namespace {

constexpr int CW = 4;  // conv width (d_conv); the FastVim configs never change it

template <int VEC>
struct ChanParams {   // per-lane conv parameters of its VEC channels
  float wf[VEC][CW], wb[VEC][CW], bf[VEC], bb[VEC];
  __device__ __forceinline__ void load(const float* wf_, const float* bf_, const float* wb_, const float* bb_, int c0, bool act) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      // one 16-byte load per channel and direction (rows of the (d_in, 4) weight are 16-byte aligned)
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (act) {
        a = *reinterpret_cast<const float4*>(wf_ + (size_t)(c0 + v) * CW);
        c = *reinterpret_cast<const float4*>(wb_ + (size_t)(c0 + v) * CW);
      }
      wf[v][0] = a.x; wf[v][1] = a.y; wf[v][2] = a.z; wf[v][3] = a.w;
      wb[v][0] = c.x; wb[v][1] = c.y; wb[v][2] = c.z; wb[v][3] = c.w;
      bf[v] = act && bf_ ? bf_[c0 + v] : 0.f;
      bb[v] = act && bb_ ? bb_[c0 + v] : 0.f;
    }
  }
};

// loads the x half of TJ+2*H tokens around tile [j0, j0+TJ) of row i (zero outside [0,L))
template <typename T, int VEC, int TJ, int H, bool TP = true>
__device__ __forceinline__ void load_x_tile(const T* xz_b, const Geo& g, int d_in, int i, int j0, int c0,
                                            bool act, float (&x)[TJ + 2 * H][VEC]) {
#pragma unroll
  for (int k = 0; k < TJ + 2 * H; ++k) {
    int s = i * g.cols + j0 - H + k;
    bool ok = act && s >= 0 && s < g.L && (j0 - H + k) < g.cols + H;
    if (ok) {
      int m = tok_mem<TP>(g, s);
      VecIO<T, VEC>::load(xz_b + (size_t)m * 2 * d_in + c0, x[k]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[k][v] = 0.f;
    }
  }
}


}  // namespace
