// split.hip -- operands of the SPLIT-bfloat16 parity path (round 4).
//
// The float32 path meets BASELINE.json's 1e-3 logit gate on `v_mfma_f32_32x32x2_f32` at 1/16 of the bf16 matrix rate.
// A float32 value is hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), so
//     x * w  =  x_hi w_hi + x_lo w_hi + x_hi w_lo  + O(2^-16 |x w|)
// and the three products are ONE bf16 GEMM over a three times longer K when the operands are concatenated along K:
//     A' = (x_hi | x_lo | x_hi),  B' = (w_hi | w_hi | w_lo)       (accumulated in float32 by the MFMA)
// For an NHWC convolution "along K" is "along the channels" (3 Ci channels per pixel, weights packed accordingly); a
// weight gradient contracts over pixels / rows instead and takes the hi and lo PLANES as three accumulating launches.
// This kernel produces those forms from a float32 matrix; everything else on the parity path stays float32.
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT_ = 256;

__device__ __forceinline__ void split1(float x, float& hi, float& lo) {
  const unsigned short h = __builtin_bit_cast(unsigned short, (__bf16)x);      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  hi = __uint_as_float((unsigned)h << 16);
  // x - hi is exact in float32 (|x - hi| <= half an ulp of hi); hi = +-Inf (x infinite, or finite but beyond bf16's range):
  // lo = 0, so that hi + lo stays Inf as on the float32 path instead of Inf - Inf = NaN
  const float d = (h & 0x7fffu) == 0x7f80u ? 0.f : x - hi;
  const unsigned short l = __builtin_bit_cast(unsigned short, (__bf16)d);
  lo = __uint_as_float((unsigned)l << 16);                                      // the bf16 value itself, also in the float32 copies
}

// vector form: cols % 8 == 0, rows of the outputs 16-byte aligned.  One thread = 8 consecutive columns of one row.
// (first-class vector types and value selects only: arrays picked through a pointer, or HIP's struct uint4 under a
// ternary, end up in scratch memory)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <bool F32OUT>
__global__ __launch_bounds__(NT_) void split_rows_kernel(const float* __restrict__ src, long long rows, int cols, long long ld_src,
                                                         void* __restrict__ cat, int order, bf16_t* __restrict__ hi_p,
                                                         bf16_t* __restrict__ lo_p) {
  const int cv = cols >> 3;
  const long long total = rows * cv;
  const bool o0 = order == 0;
  for (long long t = (long long)blockIdx.x * NT_ + threadIdx.x; t < total; t += (long long)gridDim.x * NT_) {
    const long long r = t / cv;
    const int c = (int)(t - r * cv) * 8;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src + r * ld_src + c);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(src + r * ld_src + c + 4);
    f32x4_t ha, la, hb, lb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float h, l;
      split1(a[e], h, l);
      ha[e] = h; la[e] = l;
      split1(b[e], h, l);
      hb[e] = h; lb[e] = l;
    }
    if constexpr (F32OUT) {
      float* o = reinterpret_cast<float*>(cat) + r * 3 * cols + c;
      *reinterpret_cast<f32x4_t*>(o) = ha;
      *reinterpret_cast<f32x4_t*>(o + 4) = hb;
      *reinterpret_cast<f32x4_t*>(o + cols) = o0 ? la : ha;
      *reinterpret_cast<f32x4_t*>(o + cols + 4) = o0 ? lb : hb;
      *reinterpret_cast<f32x4_t*>(o + 2 * cols) = o0 ? ha : la;
      *reinterpret_cast<f32x4_t*>(o + 2 * cols + 4) = o0 ? hb : lb;
    } else {
      const u32x4_t hv = {pack_bf16x2(ha[0], ha[1]), pack_bf16x2(ha[2], ha[3]), pack_bf16x2(hb[0], hb[1]), pack_bf16x2(hb[2], hb[3])};
      const u32x4_t lv = {pack_bf16x2(la[0], la[1]), pack_bf16x2(la[2], la[3]), pack_bf16x2(lb[0], lb[1]), pack_bf16x2(lb[2], lb[3])};
      if (cat != nullptr) {
        bf16_t* o = reinterpret_cast<bf16_t*>(cat) + r * 3 * cols + c;
        *reinterpret_cast<u32x4_t*>(o) = hv;
        *reinterpret_cast<u32x4_t*>(o + cols) = o0 ? lv : hv;
        *reinterpret_cast<u32x4_t*>(o + 2 * cols) = o0 ? hv : lv;
      }
      if (hi_p != nullptr) *reinterpret_cast<u32x4_t*>(hi_p + r * cols + c) = hv;
      if (lo_p != nullptr) *reinterpret_cast<u32x4_t*>(lo_p + r * cols + c) = lv;
    }
  }
}

// any shape (weights: small), optionally transposed: cat_t [cols][3 * rows] with block j of row c = part j of src[:, c]
template <bool F32OUT>
__global__ __launch_bounds__(NT_) void split_any_kernel(const float* __restrict__ src, long long rows, int cols, long long ld_src,
                                                        void* __restrict__ cat, int order, int transpose) {
  const long long total = rows * cols;
  for (long long t = (long long)blockIdx.x * NT_ + threadIdx.x; t < total; t += (long long)gridDim.x * NT_) {
    const long long r = t / cols;
    const int c = (int)(t - r * cols);
    float h, l;
    split1(src[r * ld_src + c], h, l);
    const float s[3] = {h, order == 0 ? l : h, order == 0 ? h : l};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const long long o = transpose ? (long long)c * 3 * rows + j * rows + r : r * 3 * cols + (long long)j * cols + c;
      if constexpr (F32OUT)
        reinterpret_cast<float*>(cat)[o] = s[j];
      else
        reinterpret_cast<bf16_t*>(cat)[o] = from_f32<bf16_t>(s[j]);
    }
  }
}

// dx [B][Hi][Wi][C] = the parity-class results of a strided conv dgrad, each a dense [B][Hq][Wq][C] float32 matrix
// (class (a, b) holds the pixels hi % sh == a, wi % sw == b), + residual
__global__ __launch_bounds__(NT_) void class_scatter_kernel(const float* __restrict__ c00, const float* __restrict__ c01,
                                                            const float* __restrict__ c10, const float* __restrict__ c11,
                                                            const float* __restrict__ residual, float* __restrict__ dx, int B,
                                                            int Hi, int Wi, int C4, int sh, int sw) {
  const long long total = (long long)B * Hi * Wi * C4;
  for (long long t = (long long)blockIdx.x * NT_ + threadIdx.x; t < total; t += (long long)gridDim.x * NT_) {
    const int c = (int)(t % C4);
    long long pix = t / C4;
    const int wi = (int)(pix % Wi);
    pix /= Wi;
    const int hi = (int)(pix % Hi);
    const int b = (int)(pix / Hi);
    const int a = hi % sh, bc = wi % sw;
    const int Hq = (Hi - a + sh - 1) / sh, Wq = (Wi - bc + sw - 1) / sw;
    const float* src = a == 0 ? (bc == 0 ? c00 : c01) : (bc == 0 ? c10 : c11);
    f32x4_t v = reinterpret_cast<const f32x4_t*>(src)[(((long long)b * Hq + hi / sh) * Wq + wi / sw) * C4 + c];
    if (residual != nullptr) v += reinterpret_cast<const f32x4_t*>(residual)[t];
    reinterpret_cast<f32x4_t*>(dx)[t] = v;
  }
}

// float32 element-wise companions of the split-bf16 Linear products (the GEMM launches write plain float32 + bias; the
// float32 path has these steps in its GEMM epilogue, same order, same rounding points): mode 0: out = gelu_erf(a)
// (timm Mlp, exact erf), 1: out = a * gelu_erf'(b), 2: out = a + b
__global__ __launch_bounds__(NT_) void ew_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                     long long n4, int mode) {
  for (long long t = (long long)blockIdx.x * NT_ + threadIdx.x; t < n4; t += (long long)gridDim.x * NT_) {
    const f32x4_t x = reinterpret_cast<const f32x4_t*>(a)[t];
    f32x4_t y = mode == 0 ? x : reinterpret_cast<const f32x4_t*>(b)[t];
    f32x4_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (mode == 0) r[e] = 0.5f * x[e] * (1.0f + erff(x[e] * 0.70710678118654752440f));
      else if (mode == 1)
        r[e] = x[e] * (0.5f * (1.0f + erff(y[e] * 0.70710678118654752440f)) + y[e] * 0.39894228040143267794f * __expf(-0.5f * y[e] * y[e]));
      else r[e] = x[e] + y[e];
    }
    reinterpret_cast<f32x4_t*>(out)[t] = r;
  }
}

}  // namespace

extern "C" int htrvt_elementwise_f32(const float* a, const float* b, float* out, int64_t n, int mode, void* stream) {
  HTRVT_REQUIRE(a != nullptr && out != nullptr && n > 0 && n % 4 == 0 && mode >= 0 && mode <= 2 && (mode == 0 || b != nullptr),
                "htrvt_elementwise_f32: bad arguments (n %% 4 == 0; mode 0 gelu(a), 1 a * gelu'(b), 2 a + b)");
  const long long n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + NT_ - 1) / NT_ > 65536 * 4 ? 65536 * 4 : (n4 + NT_ - 1) / NT_);
  hipLaunchKernelGGL(ew_f32_kernel, dim3(grid), dim3(NT_), 0, (hipStream_t)stream, a, b, out, n4, mode);
  return check_launch("elementwise_f32");
}

extern "C" int htrvt_class_scatter_f32(const float* c00, const float* c01, const float* c10, const float* c11, const float* residual,
                                       float* dx, int B, int Hi, int Wi, int C, int sh, int sw, void* stream) {
  HTRVT_REQUIRE(dx != nullptr && c00 != nullptr && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0, "htrvt_class_scatter_f32: bad shape");
  HTRVT_REQUIRE((sh == 1 || sh == 2) && (sw == 1 || sw == 2) && (sh == 1 || c10 != nullptr) && (sw == 1 || c01 != nullptr) &&
                    (sh == 1 || sw == 1 || c11 != nullptr), "htrvt_class_scatter_f32: one dense matrix per parity class");
  const long long total = (long long)B * Hi * Wi * (C / 4);
  const unsigned grid = (unsigned)((total + NT_ - 1) / NT_ > 65536 * 4 ? 65536 * 4 : (total + NT_ - 1) / NT_);
  hipLaunchKernelGGL(class_scatter_kernel, dim3(grid), dim3(NT_), 0, (hipStream_t)stream, c00, c01, c10, c11, residual, dx, B, Hi, Wi,
                     C / 4, sh, sw);
  return check_launch("class_scatter_f32");
}

extern "C" int htrvt_split_bf16(const float* src, int64_t rows, int cols, int64_t ld_src, void* cat, int order, int cat_f32,
                                int transpose, void* hi, void* lo, void* stream) {
  HTRVT_REQUIRE(src != nullptr && rows > 0 && cols > 0 && ld_src >= cols, "htrvt_split_bf16: bad source");
  HTRVT_REQUIRE(order == 0 || order == 1, "htrvt_split_bf16: order 0 = (hi, lo, hi), 1 = (hi, hi, lo)");
  HTRVT_REQUIRE(cat != nullptr || hi != nullptr || lo != nullptr, "htrvt_split_bf16: no output");
  HTRVT_REQUIRE(!(transpose || cat_f32) || (hi == nullptr && lo == nullptr && cat != nullptr),
                "htrvt_split_bf16: transposed / float32 outputs have the concatenated form only");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = !transpose && (cols % 8 == 0) && (ld_src % 4 == 0) && ((reinterpret_cast<unsigned long long>(src) & 15) == 0) &&
                   ((reinterpret_cast<unsigned long long>(cat) & 15) == 0) && ((reinterpret_cast<unsigned long long>(hi) & 15) == 0) &&
                   ((reinterpret_cast<unsigned long long>(lo) & 15) == 0);
  if (vec) {
    const long long total = (long long)rows * (cols / 8);
    const unsigned grid = (unsigned)((total + NT_ - 1) / NT_ > 65536 * 4 ? 65536 * 4 : (total + NT_ - 1) / NT_);
    if (cat_f32)
      hipLaunchKernelGGL(split_rows_kernel<true>, dim3(grid), dim3(NT_), 0, st, src, (long long)rows, cols, (long long)ld_src, cat, order,
                         (bf16_t*)nullptr, (bf16_t*)nullptr);
    else
      hipLaunchKernelGGL(split_rows_kernel<false>, dim3(grid), dim3(NT_), 0, st, src, (long long)rows, cols, (long long)ld_src, cat, order,
                         (bf16_t*)hi, (bf16_t*)lo);
    return check_launch("split_bf16");
  }
  HTRVT_REQUIRE(hi == nullptr && lo == nullptr, "htrvt_split_bf16: hi / lo planes need cols %% 8 == 0 and 16-byte aligned buffers");
  const long long total = (long long)rows * cols;
  const unsigned grid = (unsigned)((total + NT_ - 1) / NT_ > 65536 ? 65536 : (total + NT_ - 1) / NT_);
  if (cat_f32)
    hipLaunchKernelGGL(split_any_kernel<true>, dim3(grid), dim3(NT_), 0, st, src, (long long)rows, cols, (long long)ld_src, cat, order, transpose);
  else
    hipLaunchKernelGGL(split_any_kernel<false>, dim3(grid), dim3(NT_), 0, st, src, (long long)rows, cols, (long long)ld_src, cat, order, transpose);
  return check_launch("split_bf16");
}
