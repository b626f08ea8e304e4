// stem_mfma.hip -- conv1 (Cin = 1, 3x3, stride (2,1)) -> BatchNorm -> ReLU -> max-pool 3x3 / (2,1) of the stem on the
// matrix cores (reference resnet18.py:48,74-77: conv1 / bn1 / relu / maxpool), bfloat16 output.
//
// The VALU form (stem.hip stem_fused_fwd_kernel) spends 27 FMAs per pooled output on the convolution plus ~16 VALU
// operations on BatchNorm / ReLU / pooling / arg-max and runs at 2.7x its HBM floor.  Here the convolution is an MFMA
// product with the im2col built on the fly:
//     D[channel][pixel] = sum_k A[channel][k] * B[k][pixel],  v_mfma_f32_32x32x16_f16, K = 16 slots:
//       k = 0..8   the 9 taps: A = w[channel][tap] * bn_scale[channel],  B = whitened pixel (row, column) of the tap
//       k = 9, 10  the BatchNorm shift as hi + lo float16 parts:        B = 1
//       k = 11..15 zero
//   so the accumulator IS the BatchNorm output (C operand = inline 0, no accumulator initialisation).  float16 operands:
//   11-bit mantissas against the 8 of the bfloat16 result; the whitened pixels (|x| < ~10) and scaled weights are in range.
// One MFMA = 32 channels x 32 pixels of ONE conv row.  MFMA row i carries channel 16*((i>>2)&1) + (i&3) + 4*(i>>3) of its
// block of 32, which makes the 16 accumulator registers of a lane 16 CONSECUTIVE channels of one pixel: the pooled row is
// leaves the registers as 32 B of values + 16 B of arg-max bytes per lane and channel block; two channel blocks are
// gathered in a per-wave LDS stage so that the global stores are 128 B / 64 B runs per pixel (16 B pieces per lane kept the
// L2 request queues at two thirds of their rate: 419 us).
// Pooling in the accumulator layout: the three conv rows of a pooled row are three accumulator sets of the same lane
// (integer max of keys); the three columns are the neighbouring lanes (DPP wave_shr:1 / wave_shl:1).  A 32-pixel block
// produces the 30 inner pixels (blocks overlap by two columns: no carry between blocks, 6.7 % redundant MFMA work).
// Keys, ReLU and arg-max follow stem_fused_fwd_kernel's bfloat16 rule: key = float bits with the low four mantissa bits
// replaced by 8 - (3 * row + column) of the candidate, so the first maximum in scan order is a plain integer maximum,
// 0 = closed ReLU / pooling padding; idx = 3 * row + column of the arg-max, 15 where the ReLU is closed.
#include "common.h"

using namespace htrvt;

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int SM_NT = 256, SM_PIX = 30;
constexpr int SM_YROW = 144, SM_IROW = 80;                      // staged row pitch in bytes: 128 + 16, 64 + 16
constexpr int SM_STAGE = 32 * SM_YROW + 32 * SM_IROW;           // per wave

__device__ __forceinline__ float pixel_of(const void* img, long long i, int u8) {
  return u8 ? (float)reinterpret_cast<const unsigned char*>(img)[i] / 255.0f : reinterpret_cast<const float*>(img)[i];
}

__global__ __launch_bounds__(SM_NT) void stem_mfma_fwd_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                              const float* __restrict__ w, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                              unsigned char* __restrict__ idx, int H, int W, int C, int u8,
                                                              int nblk, int RS) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  _Float16* rows = reinterpret_cast<_Float16*>(smem_raw);                       // [7][RS]: LDS column = image column + 2
  f16x8_t* afr = reinterpret_cast<f16x8_t*>(smem_raw + ((7 * RS * 2 + 15) & ~15));   // [C / 32][64] A fragments
  // per-wave staging of two channel blocks (64 channels) of a 32-pixel block, so that the global stores are 128 B (values)
  // and 64 B (arg-max bytes) runs per pixel instead of 16 B pieces per lane: padded rows, conflict-free b128 accesses
  char* stage = reinterpret_cast<char*>(afr + (C / 32) * 64) + (threadIdx.x >> 6) * SM_STAGE;
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1;
  const int b = blockIdx.x / Hp, ph = blockIdx.x - b * Hp;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  for (int i = threadIdx.x; i < 7 * RS; i += SM_NT) {
    const int r = i / RS, c = i - r * RS;
    const int hi = 4 * ph - 3 + r, wi = c - 2;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (pixel_of(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
    rows[i] = (_Float16)v;
  }
  const int ncb = C / 32;
  for (int i = threadIdx.x; i < ncb * 64; i += SM_NT) {
    const int cb = i >> 6, l = i & 63, row = l & 31;
    const int ch = cb * 32 + 16 * ((row >> 2) & 1) + (row & 3) + 4 * (row >> 3);
    const float sc = scale[ch];
    f16x8_t a = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((l >> 5) == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) a[t] = (_Float16)(w[ch * 9 + t] * sc);
    } else {
      const float sh = shift[ch];
      a[0] = (_Float16)(w[ch * 9 + 8] * sc);
      a[1] = (_Float16)sh;
      a[2] = (_Float16)(sh - (float)a[1]);
    }
    afr[i] = a;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  // conv row k of this pooled row = conv row 2 ph - 1 + k; a row outside the conv output is pooling padding: its keys are
  // forced to 0 through the (block-uniform) mask and code of the v_and_or that builds them
  unsigned kmask[3], kcode[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool ok = (2 * ph - 1 + k) >= 0 && (2 * ph - 1 + k) < Hc;
    kmask[k] = ok ? ~0xFu : 0u;
    kcode[k] = ok ? (unsigned)(6 - 3 * k) : 0u;
  }
  const f32x16_t zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = wave; j < nblk; j += SM_NT / 64) {
    const int p = SM_PIX * j - 1 + n;            // this lane's pixel (B column n); outputs are the lanes n = 1 .. 30
    f16x8_t bfr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const _Float16* r0 = rows + (2 * k) * RS + p + 1;      // LDS column of pixel p - 1
      const _Float16* r1 = r0 + RS;
      const _Float16* r2 = r1 + RS;
      f16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (h == 0) {
        v[0] = r0[0], v[1] = r0[1], v[2] = r0[2];
        v[3] = r1[0], v[4] = r1[1], v[5] = r1[2];
        v[6] = r2[0], v[7] = r2[1];
      } else {
        v[0] = r2[2];
        v[1] = (_Float16)1.0f;
        v[2] = (_Float16)1.0f;
      }
      bfr[k] = v;
    }
    const bool inside = p >= 0 && p < W;
    const bool edge = j == 0 || j == nblk - 1;    // only these blocks hold columns outside the image (wave-uniform)
    for (int cb = 0; cb < ncb; ++cb) {
      const f16x8_t a = afr[cb * 64 + lane];
      const f32x16_t acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr[0], zero, 0, 0, 0);
      const f32x16_t acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr[1], zero, 0, 0, 0);
      const f32x16_t acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bfr[2], zero, 0, 0, 0);
      int m[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k0 = (int)((__float_as_uint(acc0[r]) & kmask[0]) | kcode[0]);
        const int k1 = (int)((__float_as_uint(acc1[r]) & kmask[1]) | kcode[1]);
        const int k2 = (int)((__float_as_uint(acc2[r]) & kmask[2]) | kcode[2]);
        m[r] = max(max(k0, k1), k2);
      }
      if (edge) {                                   // columns outside the image are pooling padding
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = inside ? m[r] : 0;
      }
      // Negative keys (closed ReLU) are NOT clamped here: among negative candidates the integer maximum picks an arbitrary
      // one, but any positive candidate beats them all and an all-negative window is closed either way.
      unsigned yv[8], iv[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float o2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int mm = m[r + e];
          const int left = __builtin_amdgcn_mov_dpp(mm + 2, 0x138, 0xf, 0xf, true);     // wave_shr:1: lane <- lane - 1
          const int right = __builtin_amdgcn_mov_dpp(mm, 0x130, 0xf, 0xf, true);        // wave_shl:1: lane <- lane + 1
          const int best = max(max(left, mm + 1), right);
          const bool open = best > 15;                                                  // a positive value under the code bits
          o2[e] = open ? __uint_as_float((unsigned)best & ~0xFu) : 0.f;
          const unsigned am = open ? 8u - ((unsigned)best & 15u) : 15u;
          iv[(r + e) >> 2] |= am << (8 * ((r + e) & 3));
        }
        yv[r >> 1] = pack_bf16x2(o2[0], o2[1]);
      }
      // stage this channel block (half q of the 64-channel pair)
      const int q = cb & 1;
      char* ys = stage + n * SM_YROW + q * 64 + h * 32;
      reinterpret_cast<uint4*>(ys)[0] = make_uint4(yv[0], yv[1], yv[2], yv[3]);
      reinterpret_cast<uint4*>(ys)[1] = make_uint4(yv[4], yv[5], yv[6], yv[7]);
      *reinterpret_cast<uint4*>(stage + 32 * SM_YROW + n * SM_IROW + q * 32 + h * 16) = make_uint4(iv[0], iv[1], iv[2], iv[3]);
      if (q == 1 || cb == ncb - 1) {        // pair complete (or a last single block): rows of (q + 1) * 64 B / 32 B go out
        const int cb0 = cb - q, nch = (q + 1) * 4;           // 16-byte chunks of a staged value row
        const long long row0 = (long long)blockIdx.x * W + SM_PIX * j - 1;     // pixel of staged row 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 8 * i + (lane >> 3), ch = lane & 7;
          const int pp = SM_PIX * j - 1 + row;
          if (row >= 1 && row <= SM_PIX && pp < W && ch < nch)
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y + (row0 + row) * C + cb0 * 32) + ch * 16) =
                *reinterpret_cast<const uint4*>(stage + row * SM_YROW + ch * 16);
        }
        if (idx != nullptr) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = 16 * i + (lane >> 2), ch = lane & 3;
            const int pp = SM_PIX * j - 1 + row;
            if (row >= 1 && row <= SM_PIX && pp < W && ch * 2 < nch)
              *reinterpret_cast<uint4*>(idx + (row0 + row) * C + cb0 * 32 + ch * 16) =
                  *reinterpret_cast<const uint4*>(stage + 32 * SM_YROW + row * SM_IROW + ch * 16);
          }
        }
      }
    }
  }
}

}  // namespace

namespace htrvt {

// 1 launched, 0 not served (shape outside this kernel's domain), < 0 error
int stem_mfma_try_launch(const void* img, const float* stats, const float* w, const float* scale, const float* shift, void* y,
                         uint8_t* idx, int B, int H, int W, int C, int img_u8, hipStream_t st) {
  if (C % 32 != 0 || C > 512 || H % 2 != 0 || H < 4 || W < 1) return 0;
  const int nblk = (W + SM_PIX - 1) / SM_PIX;
  const int RS = (SM_PIX * nblk + 36 + 7) & ~7;
  const size_t smem = (((size_t)7 * RS * 2 + 15) & ~(size_t)15) + (size_t)(C / 32) * 64 * 16 + (size_t)(SM_NT / 64) * SM_STAGE;
  if (smem > 160 * 1024) return 0;
  static bool attr_done = false;
  if (smem > 64 * 1024 && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_mfma_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return 0;
    attr_done = true;
  }
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1;
  hipLaunchKernelGGL(stem_mfma_fwd_kernel, dim3(B * Hp), dim3(SM_NT), smem, st, img, stats, w, scale, shift, (bf16_t*)y, idx, H, W,
                     C, img_u8, nblk, RS);
  const int rc = check_launch("stem_mfma_fwd");
  return rc ? rc : 1;
}

}  // namespace htrvt
