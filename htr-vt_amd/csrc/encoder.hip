// encoder.hip -- HBM-bound row kernels of the ViT encoder and head (reference
// HTR_VT.py:32-36 softmax, :68,75,169 nn.LayerNorm(eps 1e-6), :136,239 the
// param-free LayerNorm over all N*C logits): one wave64 per row, 16-byte
// vector loads, in-register two-pass statistics, shuffle reductions.
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT = 256;
constexpr int MAXC = 4;  // 16-byte chunks per lane kept in registers

// ------------------------------------------------------------------ LayerNorm forward (affine)
template <typename T>
__global__ __launch_bounds__(NT) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           long long rows, int D, float eps) {
  constexpr int CH = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = D / CH;
  Vec16<T> v[MAXC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      v[k].raw = reinterpret_cast<const decltype(v[k].raw)*>(x + row * D)[c];
#pragma unroll
      for (int j = 0; j < CH; ++j) s += v[k].get(j);
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float d = v[k].get(j) - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < CH; ++j)
        o.set(j, fmaf((v[k].get(j) - mean) * rstd, gamma[c * CH + j], beta[c * CH + j]));
      reinterpret_cast<decltype(o.raw)*>(y + row * D)[c] = o.raw;
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// ------------------------------------------------------------------ row softmax: float32 scores -> P (T)
template <typename T>
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows,
                                                          int n, const float* __restrict__ bias, long long bias_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = n / 4;
  float4 v[MAXC];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      v[k] = reinterpret_cast<const float4*>(s + row * n)[c];
      if (bias != nullptr) {   // additive score bias, row r of the scores uses bias row r % bias_rows ([heads][N][n] per batch entry)
        const float4 b = reinterpret_cast<const float4*>(bias + (row % bias_rows) * n)[c];
        v[k].x += b.x; v[k].y += b.y; v[k].z += b.z; v[k].w += b.w;
      }
      mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      v[k].x = __expf(v[k].x - mx);
      v[k].y = __expf(v[k].y - mx);
      v[k].z = __expf(v[k].z - mx);
      v[k].w = __expf(v[k].w - mx);
      sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      if constexpr (sizeof(T) == 4) {
        reinterpret_cast<float4*>(p + row * n)[c] = make_float4(v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv);
      } else {
        uint2 o;
        o.x = pack_bf16x2(v[k].x * inv, v[k].y * inv);
        o.y = pack_bf16x2(v[k].z * inv, v[k].w * inv);
        reinterpret_cast<uint2*>(p + row * n)[c] = o;
      }
    }
  }
}

// ------------------------------------------------------------------ param-free LN over all N*C logits of a sample
__global__ __launch_bounds__(NT) void seq_whiten_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ stats, int NC, float eps) {
  __shared__ float red[8];
  const float* xs = x + (long long)blockIdx.x * NC;
  float s = 0.f;
  for (int i = threadIdx.x; i < NC; i += NT) s += xs[i];
  const float mean = block_sum_256(s, red) / (float)NC;
  float q = 0.f;
  for (int i = threadIdx.x; i < NC; i += NT) {
    const float d = xs[i] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum_256(q, red) / (float)NC + eps);
  float* ys = y + (long long)blockIdx.x * NC;
  for (int i = threadIdx.x; i < NC; i += NT) ys[i] = (xs[i] - mean) * rstd;
  if (threadIdx.x == 0 && stats) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rstd;
  }
}

}  // namespace

extern "C" int htrvt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   int64_t rows, int D, float eps, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(D % ch == 0 && D / ch <= 64 * MAXC, "htrvt_layernorm_fwd: D=%d unsupported (multiple of %d, <= %d)", D, ch,
                64 * MAXC * ch);
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, mean, rstd, (long long)rows, D, eps);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, gamma, beta,
                       (float*)y, mean, rstd, (long long)rows, D, eps);
  return check_launch("layernorm_fwd");
}

extern "C" int htrvt_softmax_rows(const float* s, void* p, int64_t rows, int n, int dtype, const float* bias, int64_t bias_rows,
                                  void* stream) {
  HTRVT_REQUIRE(bias == nullptr || bias_rows > 0, "htrvt_softmax_rows: bias needs bias_rows > 0");
  HTRVT_REQUIRE(n % 4 == 0 && n / 4 <= 64 * MAXC, "htrvt_softmax_rows: n=%d unsupported (multiple of 4, <= %d)", n,
                64 * MAXC * 4);
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(NT), 0, (hipStream_t)stream, s, (bf16_t*)p, (long long)rows,
                       n, bias, (long long)(bias ? bias_rows : 1));
  else
    hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, s, (float*)p, (long long)rows, n,
                       bias, (long long)(bias ? bias_rows : 1));
  return check_launch("softmax_rows");
}

extern "C" int htrvt_seq_whiten_fwd(const void* x, float* y, float* stats, int B, int NC, float eps, int dtype,
                                    void* stream) {
  HTRVT_REQUIRE(dtype == HTRVT_F32, "htrvt_seq_whiten_fwd: logits are float32");
  hipLaunchKernelGGL(seq_whiten_fwd_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, (const float*)x, y, stats, NC, eps);
  return check_launch("seq_whiten_fwd");
}
