// common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "htrvt.h"

namespace htrvt {

struct bf16_t {
  unsigned short v;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((unsigned)x.v) << 16); }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short v) { return __uint_as_float(((unsigned)v) << 16); }

template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) {
  return x;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  bf16_t r;
  r.v = __builtin_bit_cast(unsigned short, b);
  return r;
}
// two floats -> one dword of two bfloat16 (RNE, NaN stays NaN).  As a VECTOR conversion: hipcc then emits ONE
// v_cvt_pk_bf16_f32; converting the halves separately and OR-ing them costs four instructions (two conversions with a
// zero partner, a shift, an SDWA or) -- and every bf16 epilogue in this library is VALU work beside its stores.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_raw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#ifdef HTRVT_EXP_OLDPACK   // A/B build of the former four-instruction form (tools: make EXTRA=-DHTRVT_EXP_OLDPACK LIB=... OBJDIR=...)
  return (unsigned)from_f32<bf16_t>(lo).v | ((unsigned)from_f32<bf16_t>(hi).v << 16);
#endif
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_raw_t));
}

// pixel i of an image batch that is float32, or uint8 taken as value / 255 (torchvision ToTensor: the uint8 -> float
// hand-off of the reference's data pipeline, data/dataset.py, fused into the first kernels that touch the image)
__device__ __forceinline__ float load_pixel(const void* img, long long i, int u8) {
  return u8 ? (float)reinterpret_cast<const unsigned char*>(img)[i] / 255.0f : reinterpret_cast<const float*>(img)[i];
}

// 16-byte vector of T
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
  __device__ __forceinline__ void set(int i, float v) { (&raw.x)[i] = v; }
};
template <>
struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    unsigned w = (&raw.x)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ void set(int i, float v) {
#ifdef HTRVT_EXP_OLDPACK
    unsigned b = from_f32<bf16_t>(v).v;
    unsigned& w = (&raw.x)[i >> 1];
    w = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
#else
    // element insertion into a first-class bfloat16 vector: set(0..7) in an unrolled loop becomes four
    // v_cvt_pk_bf16_f32 (the mask-and-or form cost a conversion, a shift and a bit-field insert per element)
    typedef __bf16 bf16x8_raw_t __attribute__((ext_vector_type(8)));
    bf16x8_raw_t h = __builtin_bit_cast(bf16x8_raw_t, raw);
    h[i] = (__bf16)v;
    raw = __builtin_bit_cast(uint4, h);
#endif
  }
};

// 16-byte streaming (non-temporal) load / store: tensors that are read or written ONCE per pass and are far larger than the L2
// (the BatchNorm apply / backward-apply passes over 100 ... 400 MB activations; htrvt_adamw's moments: 276 -> 240 us with them)
template <class RAW>
__device__ __forceinline__ RAW ld_stream16(const RAW* p) {
  typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
  static_assert(sizeof(RAW) == 16, "16-byte vectors");
  return __builtin_bit_cast(RAW, __builtin_nontemporal_load(reinterpret_cast<const u32x4s_t*>(p)));
}
template <class RAW>
__device__ __forceinline__ void st_stream16(RAW* p, const RAW& v) {
  typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));
  static_assert(sizeof(RAW) == 16, "16-byte vectors");
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4s_t, v), reinterpret_cast<u32x4s_t*>(p));
}

// tensors of at least this many bytes are streamed by the BatchNorm passes (HTRVT_BN_STREAM_MB overrides: A/B runs; 0 = always, a huge value = never)
inline long long bn_stream_bytes() {
  static const long long v = [] {
    const char* e = getenv("HTRVT_BN_STREAM_MB");
    return (e != nullptr ? atoll(e) : 150ll) << 20;
  }();
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 8 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

void set_error(const char* fmt, ...);
void set_last_kernel(const char* fmt, ...);   // symbol (as rocprofv3 prints it) of the MFMA kernel an entry point launched
int check_launch(const char* what);

}  // namespace htrvt

#define HTRVT_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      htrvt::set_error(__VA_ARGS__);    \
      return -1;                        \
    }                                   \
  } while (0)
