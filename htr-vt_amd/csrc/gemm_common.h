// gemm_common.h -- pieces shared by the two MFMA contraction kernels
// (gemm.hip: register-staged, float32 + bfloat16, any shape;
//  gemm_dma.hip: LDS-DMA staged bfloat16 256-row tiles, the throughput path).
#pragma once
#include "common.h"

namespace htrvt {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <typename T>
struct ET;
template <>
struct ET<float> {
  static constexpr int CH = 4, BK = 32, SZ = 4;
};
template <>
struct ET<bf16_t> {
  static constexpr int CH = 8, BK = 64, SZ = 2;
};

struct KParams {
  const char* A;
  const char* B;
  char* C;
  int M, N, K;
  long long lda, ldb, ldc;
  int batch_inner;
  long long sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
  int split_k, kchunk;
  long long slab_stride;      // split_k > 1: element offset of K range z's private output slab (0: all ranges share C)
  int nB, Hi, Wi, Ci, Ho, Wo, Co, kh, kw, sh, sw, ph, pw, Cpad;
  float alpha;
  int act, c_f32, accumulate;
  const float* bias;
  const float* colscale;     // per-column multiplier (eval-mode BatchNorm scale) or NULL
  char* preact;
  const char* residual;
  float* colstats;
  int tiles_m, tiles_n;
  int wo_shift, howo_shift;  // log2(Wo), log2(Ho*Wo) when powers of two, else -1 (conv-wgrad pixel decode)
  // strided conv-dgrad by input-pixel parity class: rows are the pixels (hi % sh == cls_h, wi % sw == cls_w),
  // K runs over the class's valid taps only (tapsel); cls_h < 0: all pixels, all taps
  int cls_h, cls_w, Hq, Wq, ntapsel;
  int wq_shift, hwq_shift;   // log2(Wq), log2(Hq*Wq) when both are powers of two, else -1 (epilogue row decode)
  unsigned char tapsel[12];
  // the same list, 4 bits per entry, as ONE 64-bit scalar: indexing the kernarg array with the k-tile's tap position
  // compiled to a global byte load + s_waitcnt vmcnt(0) inside the loader waves' issue path, twice per k-tile
  unsigned long long tappack;
  // HtrvtGemmDesc.A2: byte distance from A to the second gathered tensor, read through tap code kh*kw (the 1x1
  // downsample gradient as one more tap of the class-(0,0) launch); 0 without one
  unsigned extra_off;
  // fused backward-of-ReLU and BatchNorm-backward column sums in the bf16 staged epilogue (conv dgrad outputs)
  const char* relu_src;
  int relu_bits;              // relu_src is a bit mask: bit (m * ldc + n) & 7 of byte (m * ldc + n) >> 3 (HtrvtGemmDesc.relu_bits)
  const char* bnb_x[2];
  const float* bnb_mean[2];
  const float* bnb_rstd[2];
  float* bnb_partial[2];
  int bnb_tile0;
  // ReLU mask recomputed from bnb_x[0] (HtrvtGemmDesc.relu_scale / relu_shift), or NULL
  const float* relu_sc;
  const float* relu_sf;
};

// Few-row Linear launches (the per-rank batches of a strong-scaling run: M = 4096 at 16 images per GPU): with N <= 1024 a
// 256 x 192 tiling leaves three quarters of the CUs idle (16 x 4 = 64 workgroups); 256 x 128 tiles with three LDS stages
// give 1.5x the workgroups and measured +20...30 % (tools/bench_gemm.py --only encsmall --tiles 0 8: fc2 forward 57 -> 42 us,
// fc1 dgrad 53 -> 40, qkv dgrad 42 -> 33 at M = 4096; +15...25 % at M = 8192; wide N and M = 32768 lose, so they stay)
inline bool gemm_small_m_prefers_bn128(const HtrvtGemmDesc* d) {
  if (d->tile != 0 || d->gather != HTRVT_GATHER_NONE || d->dtype != HTRVT_BF16) return false;
  if (d->a_layout != HTRVT_KMAJOR || d->b_layout != HTRVT_KMAJOR || d->batch > 1 || d->split_k > 1) return false;
  if (d->N > 1024 || d->N % 128 != 0) return false;
  const long long tiles192 = (long long)((d->M + 255) / 256) * ((d->N + 191) / 192);
  return tiles192 <= 128;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// GELU for results that are rounded to bfloat16 anyway, two elements at a time so that the arithmetic runs on the packed
// float32 pipes (v_pk_fma_f32 / v_pk_mul_f32: the GELU epilogue of fc1 is VALU-bound, 100 M elements per launch).
//   erfc(|x| / sqrt 2) = exp2(a * R(a)),  a = min(|x|, 4 sqrt 2),  R of degree 5 (weighted least squares on Chebyshev nodes,
//   tools/fit_gelu.py): |erf error| < 3.5e-7 in float32, no reciprocal, ONE v_exp_f32 (libm's erff branches; the former
//   Abramowitz-Stegun form took a v_rcp_f32 and a v_exp_f32, quarter-rate each).
//   gelu(x)  = x Phi(x)  = 0.5 x + 0.5 |x| (1 - e)
//   gelu'(x) = Phi(x) + x pdf(x) = 0.5 + copysign(0.5 - 0.5 e, x) + x / sqrt(2 pi) * exp2(-x^2 log2(e) / 2)
// Measured against float64 over [-12, 12] and N(0, 4) samples: |gelu error| < 5.2e-7, |gelu' error| < 2.7e-7.
__device__ __forceinline__ f32x2_t splat2(float c) { return f32x2_t{c, c}; }
__device__ __forceinline__ f32x2_t erfc_abs2(f32x2_t ax) {
  f32x2_t a;
  a.x = __builtin_amdgcn_fmed3f(ax.x, 0.f, 5.656854249f);
  a.y = __builtin_amdgcn_fmed3f(ax.y, 0.f, 5.656854249f);
  f32x2_t r = __builtin_elementwise_fma(splat2(1.986000183e-05f), a, splat2(-6.623090474e-04f));
  r = __builtin_elementwise_fma(r, a, splat2(7.759703627e-03f));
  r = __builtin_elementwise_fma(r, a, splat2(-5.296444113e-02f));
  r = __builtin_elementwise_fma(r, a, splat2(-4.590662242e-01f));
  r = __builtin_elementwise_fma(r, a, splat2(-1.151119094e+00f));
  const f32x2_t t = a * r;
  return f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
}
__device__ __forceinline__ f32x2_t gelu_erf_fast2(f32x2_t x) {
  const f32x2_t ax = __builtin_elementwise_abs(x);
  const f32x2_t e = erfc_abs2(ax);
  const f32x2_t h = ax * splat2(0.5f);
  const f32x2_t m = __builtin_elementwise_fma(x, splat2(0.5f), h);
  return __builtin_elementwise_fma(-h, e, m);
}
__device__ __forceinline__ f32x2_t gelu_erf_grad_fast2(f32x2_t x) {
  const f32x2_t e = erfc_abs2(__builtin_elementwise_abs(x));
  const f32x2_t d = __builtin_elementwise_copysign(__builtin_elementwise_fma(e, splat2(-0.5f), splat2(0.5f)), x);
  const f32x2_t q = (x * x) * splat2(-0.72134752044448170368f);
  const f32x2_t pdf = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  return __builtin_elementwise_fma(x * splat2(0.39894228040143267794f), pdf, splat2(0.5f)) + d;
}

// Epilogue of one wave's TM x TN block of 32x32 accumulator tiles.
//   acc[i][j][r] is C(row = mrow0 + 32 i + (r&3) + 8 (r>>2) + 4 (lane>>5), col = ncol0 + 32 j + (lane&31))
// NWM = number of waves stacked along M in the workgroup (for the BN column-sum reduction through LDS).
// EXTRAS: compile the per-column scale and the trailing ReLU (eval-mode BatchNorm folded into a convolution); the
// 256-column LDS-DMA kernels (float32 split-K weight gradients) leave them out: they have no SGPRs to spare
// P: KParams, or KParams in the constant address space (the LDS-DMA kernels read it straight from the kernarg segment)
template <typename T, int TM, int TN, int NWM, int BN, int NTHREADS, bool EXTRAS = true, class P = KParams>
__device__ __forceinline__ void gemm_epilogue(f32x16_t (&acc)[TM][TN], const P& p, char* Cb, long long coff, int mrow0,
                                              int ncol0, int wm, int n0, int tile_m, int lane, char* smem,
                                              bool active = true) {
  // active == false: a wave that holds no accumulators (dedicated loader wave); it only joins the barriers
  const int h = lane >> 5, cl = lane & 31;
  if (!active) mrow0 = 0x3fffff00;  // every row test below fails
  float cs1[TN], cs2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs1[j] = cs2[j] = 0.f;

  // Fast path (wave-uniform): float32 C, the wave's whole TM x TN block inside the matrix, nothing but alpha / bias /
  // store-or-atomic-add -- attention scores, weight-gradient split-K.  No per-element predicates or branches.
  if (active && p.c_f32 && p.act == 0 && p.preact == nullptr && p.residual == nullptr && p.colstats == nullptr &&
      mrow0 + TM * 32 <= p.M && ncol0 + TN * 32 <= p.N) {
    float* Cf = reinterpret_cast<float*>(Cb) + coff;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = ncol0 + j * 32 + cl;
        const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
        const float al = (EXTRAS && p.colscale != nullptr) ? p.alpha * p.colscale[n] : p.alpha;
        float* col = Cf + (long long)(mrow0 + i * 32 + 4 * h) * p.ldc + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r] * al + bias;
          float* dst = col + (long long)((r & 3) + 8 * (r >> 2)) * p.ldc;
          if (p.accumulate)
            atomicAdd(dst, v);
          else
            *dst = v;
        }
      }
    }
    return;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = ncol0 + j * 32 + cl;
      const bool nok = n < p.N;
      const float bias = (p.bias != nullptr && nok) ? p.bias[n] : 0.f;
      const float al = (EXTRAS && p.colscale != nullptr && nok) ? p.alpha * p.colscale[n] : p.alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M && nok) {
          const float a = acc[i][j][r];
          cs1[j] += a;
          cs2[j] += a * a;
          float v = a * al + bias;
          const long long o = coff + (long long)m * p.ldc + n;
          if (p.act == 2) {  // backward of GELU: multiply by gelu'(saved pre-activation)
            const float xp = p.c_f32 ? reinterpret_cast<const float*>(p.preact)[o]
                                     : to_f32(reinterpret_cast<const T*>(p.preact)[o]);
            v *= gelu_erf_grad(xp);
          } else if (p.preact != nullptr) {
            if (p.c_f32)
              reinterpret_cast<float*>(p.preact)[o] = v;
            else
              reinterpret_cast<T*>(p.preact)[o] = from_f32<T>(v);
          }
          if (p.act == 1) v = gelu_erf(v);
          if (p.c_f32) {
            if (p.residual != nullptr) v += reinterpret_cast<const float*>(p.residual)[o];
            if (EXTRAS && p.act == 3) v = fmaxf(v, 0.f);
            if (p.accumulate)
              atomicAdd(reinterpret_cast<float*>(Cb) + o, v);
            else
              reinterpret_cast<float*>(Cb)[o] = v;
          } else {
            if (p.residual != nullptr) v += to_f32(reinterpret_cast<const T*>(p.residual)[o]);
            if (EXTRAS && p.act == 3) v = fmaxf(v, 0.f);
            reinterpret_cast<T*>(Cb)[o] = from_f32<T>(v);
          }
        }
      }
    }
  }

  if (p.colstats != nullptr) {  // per-M-tile column sums for train-mode BatchNorm (workgroup-uniform branch)
    float* red = reinterpret_cast<float*>(smem);  // [NWM][BN][2]; the operand tiles are dead by now
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = cs1[j] + __shfl_xor(cs1[j], 32, 64);
      const float s2 = cs2[j] + __shfl_xor(cs2[j], 32, 64);
      if (h == 0 && active) {
        const int c = ncol0 - n0 + j * 32 + cl;
        red[(wm * BN + c) * 2 + 0] = s1;
        red[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < BN; c += NTHREADS) {
      const int n = n0 + c;
      if (n < p.N) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) {
          a += red[(w * BN + c) * 2];
          q += red[(w * BN + c) * 2 + 1];
        }
        float* dst = p.colstats + (long long)tile_m * 2 * p.N;
        dst[n] = a;
        dst[p.N + n] = q;
      }
    }
  }
}

// defined in gemm_dma.hip: returns 1 if it launched, 0 if the shape is not one it serves, <0 on error
int gemm_dma_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st);
int gemm_dma_num_mtiles(const HtrvtGemmDesc* d);
// defined in gemm8p.hip (8-phase kernels, 256-row tiles): same return convention
int gemm8p_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st);
int gemm8pt_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st);   // MN-major x MN-major, float32 C
bool gemm8pt_serves(const HtrvtGemmDesc* d);

}  // namespace htrvt
