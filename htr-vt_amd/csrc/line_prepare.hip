// line_prepare.hip -- the step in front of the path: a ragged batch of grey uint8 scans -> the model's
// [B,1,H,W] uint8 batch, i.e. what /root/reference/data/dataset.py:104-135 does per image on the host
//   npThum      : width' = min(int(w * H / h), W);  PIL.Image.resize((width', H))      (Pillow, BICUBIC for mode 'L')
//   get_images  : img_as_float32 (value / 255) and right pad with 1.0 up to W
// bit for bit: Pillow's 8-bit resampler (src/libImaging/Resample.c; reference pin pillow==10.3.0) is integer arithmetic
// on double-precision coefficient tables -- bicubic (a = -0.5) weights over a support of 2 * max(scale, 1) source pixels,
// normalised, rounded to 22-bit fixed point, a horizontal pass and a vertical pass over its uint8 result, each
// out = clip((2^21 + sum pixel * coeff) >> 22, 0, 255).  The pad value 1.0 is the byte 255; the model's first kernels
// read the bytes as value / 255 (htrvt_img_stats / htrvt_conv1_fwd, img_u8 = 1), so no float image ever exists.
//
// HBM-bound byte work: one thread per output pixel column, coefficients built once per thread (horizontal pass) or once
// per block (vertical pass) in double precision with FMA contraction off (the x86-64 build of Pillow has none), then
// reused down the rows; neighbouring threads read neighbouring bytes.
#include "common.h"

#pragma clang fp contract(off)

using namespace htrvt;

namespace {

constexpr int NT = 256;
constexpr int PREC = 32 - 8 - 2;     // Resample.c PRECISION_BITS
constexpr int KSMAX = 48;            // taps per output pixel: ksize = 2 * ceil(2 * scale) + 1 -> scale <= 11.5
constexpr int RPB = 32;              // source rows per block of the horizontal pass

__device__ __forceinline__ double bicubic(double x) {   // Resample.c bicubic_filter, a = -0.5
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((-0.5 + 2.0) * x - (-0.5 + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * -0.5;
  return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output index xx; returns the tap count, kk[0..count)
__device__ int coeffs(int in_size, int out_size, int xx, int* kk, int kstride, int& xmin_out) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > KSMAX) xmax = KSMAX;   // never past the tap table (over-scale images are blanked by the callers, see over_scale)
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic((x + xmin - center + 0.5) * ss);
  for (int x = 0; x < xmax; ++x) {
    double w = bicubic((x + xmin - center + 0.5) * ss);
    if (ww != 0.0) w /= ww;
    kk[x * kstride] = w < 0 ? (int)(-0.5 + w * (1 << PREC)) : (int)(0.5 + w * (1 << PREC));
  }
  xmin_out = xmin;
  return xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PREC;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ int thumb_width(int h, int w, int H, int W) {   // npThum: min(int(w * H / h), W)
  const int y = (int)((double)((long long)w * H) / (double)h);
  return y < W ? y : W;
}

// An image that shrinks by more than htrvt_line_max_scale() needs more taps than the LDS tables hold.  The table lives in
// device memory, so the host entry point cannot reject it: such an image (and a degenerate one: h, w or width' < 1) is
// rendered as an empty line -- all 255, the pad value -- instead of overrunning the tables.  htrvt_amd.prepare_lines
// raises before it gets here.
__device__ __forceinline__ bool over_scale(int h, int w, int H, int ow) {
  constexpr int MAXS = (KSMAX - 1) / 4;
  return h < 1 || w < 1 || ow < 1 || h > MAXS * H || w > MAXS * ow;
}

// horizontal pass: tmp[row][xx] for row < h, xx < width'
__global__ __launch_bounds__(NT) void line_hpass_kernel(const unsigned char* __restrict__ src,
                                                        const HtrvtLineImage* __restrict__ table,
                                                        unsigned char* __restrict__ tmp, int H, int W) {
  __shared__ int kk[KSMAX * NT];     // [tap][thread]: conflict-free
  const HtrvtLineImage e = table[blockIdx.z];
  const int row0 = blockIdx.y * RPB;
  if (row0 >= e.h || e.h < 1) return;
  const int ow = thumb_width(e.h, e.w, H, W);
  const int xx = blockIdx.x * NT + threadIdx.x;
  if (blockIdx.x * NT >= ow || over_scale(e.h, e.w, H, ow)) return;
  const unsigned char* s = src + e.src_offset;
  unsigned char* t = tmp + e.tmp_offset;
  const int row1 = min(e.h, row0 + RPB);
  if (ow == e.w) {                    // Pillow skips a pass that does not change the size
    if (xx < ow)
      for (int r = row0; r < row1; ++r) t[(long long)r * W + xx] = s[(long long)r * e.w + xx];
    return;
  }
  int xmin = 0, cnt = 0;
  if (xx < ow) cnt = coeffs(e.w, ow, xx, kk + threadIdx.x, NT, xmin);
  if (xx >= ow) return;
  for (int r = row0; r < row1; ++r) {
    const unsigned char* sr = s + (long long)r * e.w + xmin;
    int acc = 1 << (PREC - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)sr[x] * kk[x * NT + threadIdx.x];
    t[(long long)r * W + xx] = clip8(acc);
  }
}

// vertical pass + right pad: dst[b][yy][x]
__global__ __launch_bounds__(NT) void line_vpass_kernel(const HtrvtLineImage* __restrict__ table,
                                                        const unsigned char* __restrict__ tmp,
                                                        unsigned char* __restrict__ dst, int H, int W) {
  extern __shared__ int vk[];        // [H][KSMAX + 2]: xmin, count, taps
  const HtrvtLineImage e = table[blockIdx.z];
  int ow = e.h >= 1 ? thumb_width(e.h, e.w, H, W) : 0;
  if (over_scale(e.h, e.w, H, ow)) ow = 0;      // blank line
  const int x = blockIdx.x * NT + threadIdx.x;
  unsigned char* d = dst + (long long)blockIdx.z * H * W;
  if (blockIdx.x * NT >= ow) {       // the whole block is padding
    if (x < W)
      for (int yy = 0; yy < H; ++yy) d[(long long)yy * W + x] = 255;
    return;
  }
  const unsigned char* t = tmp + e.tmp_offset;
  const bool resize = e.h != H;
  if (resize) {
    for (int yy = threadIdx.x; yy < H; yy += NT) {
      int* k = vk + yy * (KSMAX + 2);
      int xmin;
      k[1] = coeffs(e.h, H, yy, k + 2, 1, xmin);
      k[0] = xmin;
    }
    __syncthreads();
  }
  if (x >= W) return;
  for (int yy = 0; yy < H; ++yy) {
    unsigned char v = 255;           // 1.0: the pad of get_images (dataset.py:129-130)
    if (x < ow) {
      if (resize) {
        const int* k = vk + yy * (KSMAX + 2);
        const int ymin = k[0], cnt = k[1];
        int acc = 1 << (PREC - 1);
        for (int y = 0; y < cnt; ++y) acc += (int)t[(long long)(ymin + y) * W + x] * k[2 + y];
        v = clip8(acc);
      } else {
        v = t[(long long)yy * W + x];
      }
    }
    d[(long long)yy * W + x] = v;
  }
}

}  // namespace

extern "C" int htrvt_line_max_scale(void) { return (KSMAX - 1) / 4; }

extern "C" int htrvt_line_prepare(const uint8_t* src, const HtrvtLineImage* table, uint8_t* tmp, uint8_t* dst, int B, int H,
                                  int W, int max_src_h, void* stream) {
  HTRVT_REQUIRE(src && table && tmp && dst, "htrvt_line_prepare: null argument");
  HTRVT_REQUIRE(B > 0 && H > 0 && H <= 1024 && W > 0 && max_src_h > 0, "htrvt_line_prepare: bad shape B=%d H=%d W=%d", B, H, W);
  hipStream_t st = (hipStream_t)stream;
  const dim3 gh((W + NT - 1) / NT, (max_src_h + RPB - 1) / RPB, B);
  hipLaunchKernelGGL(line_hpass_kernel, gh, dim3(NT), 0, st, src, table, tmp, H, W);
  const dim3 gv((W + NT - 1) / NT, 1, B);
  hipLaunchKernelGGL(line_vpass_kernel, gv, dim3(NT), (size_t)H * (KSMAX + 2) * sizeof(int), st, table, (const unsigned char*)tmp, dst, H, W);
  return check_launch("line_prepare");
}
