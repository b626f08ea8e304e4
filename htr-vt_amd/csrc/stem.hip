// stem.hip -- HBM-bound kernels of the CNN stem (reference resnet18.py:42-84,
// HTR_VT.py:134-136,224-227): image whitening statistics, the Cin=1 first
// convolution, train/eval BatchNorm coefficient kernels, BN-apply/ReLU/residual,
// BN+ReLU+max-pool, and the final pool -> token assembly.  Activations are NHWC
// so every kernel streams 16-byte channel vectors (coalesced, wave64).
#include "common.h"

using namespace htrvt;

namespace htrvt {
int stem_mfma_try_launch(const void* img, const float* stats, const float* w, const float* scale, const float* shift, void* y,
                         uint8_t* idx, int B, int H, int W, int C, int img_u8, hipStream_t st);
}

namespace {

constexpr int NT = 256;

// HTRVT_STEM_VALU=1: keep the float32-FMA stem kernel on the bfloat16 path too (A/B runs, tests of that kernel)
bool stem_force_valu() {
  static const bool v = [] {
    const char* e = getenv("HTRVT_STEM_VALU");
    return e != nullptr && e[0] == '1';
  }();
  return v;
}

// ------------------------------------------------------------------ img_stats
// one block of 1024 threads per image; two passes (mean, then centred variance) -> {mean, rstd}; four 16-byte loads in
// flight per thread (256 threads with one load in flight each were a chain of 2 x 64 round trips: 53 us for 34 MB)
constexpr int NT_IMG = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float* red) {   // red: 16 floats of LDS; fixed summation order
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < NT_IMG / 64; ++k) t += red[k];
  return t;
}

__global__ __launch_bounds__(NT_IMG) void img_stats_kernel(const void* __restrict__ img, float* __restrict__ stats, int HW,
                                                           float eps, int u8) {
  __shared__ float red[NT_IMG / 64];
  const long long base = (long long)blockIdx.x * HW;
  auto load4 = [&](int i, float (&v)[4]) {
    if (u8) {
      const uchar4 q = *reinterpret_cast<const uchar4*>(reinterpret_cast<const unsigned char*>(img) + base + i);
      v[0] = (float)q.x / 255.0f;
      v[1] = (float)q.y / 255.0f;
      v[2] = (float)q.z / 255.0f;
      v[3] = (float)q.w / 255.0f;
    } else {
      const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(img) + base + i);
      v[0] = q.x;
      v[1] = q.y;
      v[2] = q.z;
      v[3] = q.w;
    }
  };
  constexpr int STEP = NT_IMG * 4, U = 4;
  float s = 0.f;
  int i = threadIdx.x * 4;
  for (; i + (U - 1) * STEP < HW; i += U * STEP) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) load4(i + u * STEP, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  for (; i < HW; i += STEP) {
    float v[4];
    load4(i, v);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = block_sum_1024(s, red) / (float)HW;
  float q = 0.f;
  i = threadIdx.x * 4;
  for (; i + (U - 1) * STEP < HW; i += U * STEP) {
    float v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) load4(i + u * STEP, v[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a = v[u][0] - mean, b = v[u][1] - mean, c = v[u][2] - mean, d = v[u][3] - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  for (; i < HW; i += STEP) {
    float v[4];
    load4(i, v);
    const float a = v[0] - mean, b = v[1] - mean, c = v[2] - mean, d = v[3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float var = block_sum_1024(q, red) / (float)HW;
  if (threadIdx.x == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rsqrtf(var + eps);
  }
}

// ------------------------------------------------------------------ conv1 fwd
// one block per output row (b, ho); the 3 whitened input rows live in LDS (zero
// halo = zero padding in whitened space); each thread owns CH consecutive output
// channels (weights in registers) and walks the row's pixels.
template <typename T>
__global__ __launch_bounds__(NT) void conv1_fwd_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                       const float* __restrict__ w, T* __restrict__ out,
                                                       float* __restrict__ colstats, int H, int W, int C, int nthr, int u8) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rows = reinterpret_cast<float*>(smem_raw);  // [3][W+2]
  const int Ho = H / 2;
  const int b = blockIdx.x / Ho, ho = blockIdx.x - b * Ho;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const int WP = W + 2;
  for (int i = threadIdx.x; i < 3 * WP; i += NT) {
    const int r = i / WP, c = i - r * WP;
    const int hi = 2 * ho - 1 + r, wi = c - 1;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (load_pixel(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
    rows[i] = v;
  }
  __syncthreads();
  const int lanes_per_pix = C / CH;
  const int ppb = nthr / lanes_per_pix;
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
  if ((int)threadIdx.x < nthr) {
    const int cg = threadIdx.x % lanes_per_pix, p0 = threadIdx.x / lanes_per_pix;
    float wr[CH][9];
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[j][t] = w[(cg * CH + j) * 9 + t];
    T* orow = out + ((long long)blockIdx.x * W) * C + cg * CH;
    for (int px = p0; px < W; px += ppb) {
      float xin[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) xin[r * 3 + c] = rows[r * WP + px + c];
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(wr[j][t], xin[t], a);
        s1[j] += a;
        s2[j] += a * a;
        o.set(j, a);
      }
      *reinterpret_cast<decltype(o.raw)*>(orow + (long long)px * C) = o.raw;
    }
  }
  // per-channel partial sums of this row: reduce the ppb threads that share a channel group
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem_raw);  // [nthr][2*CH]
  if ((int)threadIdx.x < nthr) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      red[threadIdx.x * 2 * CH + j] = s1[j];
      red[threadIdx.x * 2 * CH + CH + j] = s2[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    const int cg = c / CH, j = c - cg * CH;
    float a = 0.f, q = 0.f;
    for (int k = 0; k < ppb; ++k) {
      a += red[(k * lanes_per_pix + cg) * 2 * CH + j];
      q += red[(k * lanes_per_pix + cg) * 2 * CH + CH + j];
    }
    colstats[(long long)blockIdx.x * 2 * C + c] = a;
    colstats[(long long)blockIdx.x * 2 * C + C + c] = q;
  }
}

// ------------------------------------------------------------------ fused stem forward
// conv1 (ONE input channel) -> train-mode BatchNorm -> ReLU -> max_pool2d(3, stride (2,1), pad 1) without ever writing
// the conv1 tensor (1.6 GB at B=128, 64x1024, C1=192 in bf16; it was written once and read once).
//
// Batch statistics without the tensor: with xw_t(pos) the whitened image tap t at conv position pos,
//   y_c(pos) = sum_t W[c][t] xw_t(pos)   =>   sum_pos y_c   = sum_t W[c][t] X[t]
//                                              sum_pos y_c^2 = sum_{t,u} W[c][t] W[c][u] R[t][u]
// where X[t] = sum_pos xw_t and R[t][u] = sum_pos xw_t xw_u are 9 + 45 sums over the IMAGE (stem_moments_kernel: one
// pass over 8-34 MB), combined per channel in double precision (stem_stats_kernel) into the (sum, sum of squares) row
// that htrvt_bn_finalize takes.  conv1_bwd.hip uses the same identities for the backward.
constexpr int NMOM = 54;   // 9 tap sums + 45 upper-triangle tap products

// one block per MOM_R consecutive conv rows of one image: partial[block][64] (54 used); the 54 block reductions are
// amortised over MOM_R rows
constexpr int MOM_R = 4;
__global__ __launch_bounds__(NT) void stem_moments_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                          float* __restrict__ partial, int H, int W, int u8) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rows = reinterpret_cast<float*>(smem_raw);  // [2 MOM_R + 1][W+2]
  __shared__ float red[NT / 64][NMOM];
  const int Ho = H / 2, groups = (Ho + MOM_R - 1) / MOM_R;
  const int b = blockIdx.x / groups, ho0 = (blockIdx.x - b * groups) * MOM_R;
  const int nr = min(MOM_R, Ho - ho0);
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const int WP = W + 2;
  for (int i = threadIdx.x; i < (2 * nr + 1) * WP; i += NT) {
    const int r = i / WP, c = i - r * WP;
    const int hi = 2 * ho0 - 1 + r, wi = c - 1;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (load_pixel(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
    rows[i] = v;
  }
  __syncthreads();
  float acc[NMOM];
#pragma unroll
  for (int k = 0; k < NMOM; ++k) acc[k] = 0.f;
  for (int q = 0; q < nr; ++q)
  for (int px = threadIdx.x; px < W; px += NT) {
    float x[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) x[r * 3 + c] = rows[(2 * q + r) * WP + px + c];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t] += x[t];
#pragma unroll
      for (int u = t; u < 9; ++u) acc[9 + t * 9 - t * (t - 1) / 2 + (u - t)] += x[t] * x[u];   // upper triangle, row-major
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NMOM; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float v = 0.f;
    if (threadIdx.x < NMOM) {
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) v += red[w][threadIdx.x];
    }
    partial[(long long)blockIdx.x * 64 + threadIdx.x] = v;
  }
}

// one block of 512 threads: 64 columns x 8 row lanes sum the partial rows in double, then thread c < C forms the
// channel's (sum y, sum y^2) -> colstats[0][c], colstats[1][c]
constexpr int SS_RL = 8;
__global__ __launch_bounds__(64 * SS_RL) void stem_stats_kernel(const float* __restrict__ partial, int nrows,
                                                                const float* __restrict__ w, float* __restrict__ colstats, int C) {
  __shared__ double red[SS_RL][64];
  __shared__ double X[9], R[9][9];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  double a = 0.0;
  int r = rl;
  for (; r + 7 * SS_RL < nrows; r += 8 * SS_RL) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(r + SS_RL * u) * 64 + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; r < nrows; r += SS_RL) a += partial[(long long)r * 64 + col];
  red[rl][col] = a;
  __syncthreads();
  if (threadIdx.x < NMOM) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < SS_RL; ++k) t += red[k][threadIdx.x];
    if (threadIdx.x < 9) {
      X[threadIdx.x] = t;
    } else {
      int k = threadIdx.x - 9, i = 0;
      while (k >= 9 - i) {
        k -= 9 - i;
        ++i;
      }
      R[i][i + k] = t;
      R[i + k][i] = t;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 64 * SS_RL) {
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < 9; ++t) {
      const double wt = w[c * 9 + t];
      s1 += wt * X[t];
      double rr = 0.0;
      for (int u = 0; u < 9; ++u) rr += (double)w[c * 9 + u] * R[t][u];
      s2 += wt * rr;
    }
    colstats[c] = (float)s1;
    colstats[C + c] = (float)s2;
  }
}

// One block per pooled row (b, ph): the 7 whitened image rows that its 3 conv rows touch live in LDS.  A thread owns CH
// consecutive channels (weights, scale, shift in registers) and a contiguous run of columns: per new column it computes
// the 3 conv rows (same FMA order as conv1_fwd_kernel), BatchNorm + ReLU, the column's maximum with its row, and
// combines it with the two previous columns into the pooled output -- first maximum in (row, column) scan order, as
// ATen's max_pool2d and bn_relu_maxpool_kernel.  idx = 3 * row + column of the arg-max, 15 where the ReLU is closed.
template <typename T>
__global__ __launch_bounds__(NT) void stem_fused_fwd_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                            const float* __restrict__ w, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ y,
                                                            unsigned char* __restrict__ idx, int H, int W, int C, int nthr,
                                                            int u8) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rows = reinterpret_cast<float*>(smem_raw);  // [7][W+2]
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1, WP = W + 2;
  const int b = blockIdx.x / Hp, ph = blockIdx.x - b * Hp;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  for (int i = threadIdx.x; i < 7 * WP; i += NT) {
    const int r = i / WP, c = i - r * WP;
    const int hi = 4 * ph - 3 + r, wi = c - 1;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (load_pixel(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
    rows[i] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x >= nthr) return;
  const int lanes_per_pix = C / CH, ppb = nthr / lanes_per_pix;
  const int cg = threadIdx.x % lanes_per_pix, pl = threadIdx.x / lanes_per_pix;
  const int seg = (W + ppb - 1) / ppb, w0 = pl * seg, w1 = min(W, w0 + seg);
  if (w0 >= w1) return;
  // channel pairs as 2-vectors: the 27 FMAs of a (column, channel) become v_pk_fma_f32 over two channels (same
  // per-channel operation order as conv1_fwd_kernel, so the float32 values are identical)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr bool KEYED = sizeof(T) == 2;   // bfloat16 output: see below
  f32x2 wr[CH / 2][9], sc[CH / 2], sf[CH / 2];
#pragma unroll
  for (int j = 0; j < CH / 2; ++j) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = f32x2{w[(cg * CH + 2 * j) * 9 + t], w[(cg * CH + 2 * j + 1) * 9 + t]};
    sc[j] = f32x2{scale[cg * CH + 2 * j], scale[cg * CH + 2 * j + 1]};
    sf[j] = f32x2{shift[cg * CH + 2 * j], shift[cg * CH + 2 * j + 1]};
    if constexpr (KEYED) {   // the BatchNorm scale rides in the weights, the shift is the accumulator's start value
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[j][t] *= sc[j];
    }
  }
  // conv row k of this pooled row = conv row 2 ph - 1 + k; rows outside the conv output are pooling padding (-inf)
  bool rowok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) rowok[k] = (2 * ph - 1 + k) >= 0 && (2 * ph - 1 + k) < Hc;
  // the two previous columns' (maximum, its row)
  float pv[2][CH];
  int pr[2][CH];
  // KEYED (bfloat16 output): a candidate is ONE signed key -- the float32 bits of the pre-ReLU value with the low four
  // mantissa bits replaced by (2 - row) << 2 | (2 - column) -- so the first maximum in scan order is a plain integer max
  // (v_max3_i32) instead of compare + select chains, and the ReLU is the key 0 every maximum starts from (negative
  // floats are negative integers).  Values closer than 16 float32 ulps count as equal, which the bfloat16 result
  // cannot see.  The float32 path keeps the exact rule.
  int pk[2][CH];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      pv[q][j] = -INFINITY;
      pr[q][j] = 0;
      pk[q][j] = 0;
    }
  T* yrow = y + ((long long)blockIdx.x * W) * C + cg * CH;
  unsigned char* irow = idx ? idx + ((long long)blockIdx.x * W) * C + cg * CH : nullptr;
  for (int c = w0 - 1; c <= w1; ++c) {   // conv column c; the output of column c - 1 is complete once c is known
    float cv[CH];
    int cr[CH];
    int ck[CH];
    if (c >= 0 && c < W) {
      float xin[7][3];
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[r][d] = rows[r * WP + c + d];
#pragma unroll
      for (int j = 0; j < CH / 2; ++j) {
        float m[2] = {-INFINITY, -INFINITY};
        int mr[2] = {0, 0};
        int mk[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (!rowok[k]) continue;   // block-uniform: pooling padding above the first / below the last conv row
          f32x2 a = KEYED ? sf[j] : f32x2{0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float xv = xin[2 * k + t / 3][t % 3];
            a = __builtin_elementwise_fma(wr[j][t], f32x2{xv, xv}, a);
          }
          const f32x2 bn = KEYED ? a : __builtin_elementwise_fma(a, sc[j], sf[j]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if constexpr (KEYED) {
              mk[e] = max(mk[e], (int)((__float_as_uint(bn[e]) & ~0xFu) | ((2u - k) << 2)));
            } else {
              const float v = fmaxf(bn[e], 0.f);
              const bool take = v > m[e];
              m[e] = take ? v : m[e];
              mr[e] = take ? k : mr[e];
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          cv[2 * j + e] = m[e];
          cr[2 * j + e] = mr[e];
          ck[2 * j + e] = mk[e];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        cv[j] = -INFINITY;
        cr[j] = 0;
        ck[j] = 0;
      }
    }
    const int wo = c - 1;
    if (wo >= w0 && wo < w1) {   // window columns wo-1, wo, wo+1 = pv[0], pv[1], cv
      Vec16<T> o;
      unsigned am[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if constexpr (KEYED) {
          const unsigned best = (unsigned)max(max(pk[0][j] | 2, pk[1][j] | 1), ck[j]);   // >= 0: the ReLU
          const float mv = __uint_as_float(best & ~0xFu);
          o.set(j, mv);
          am[j] = !(mv > 0.f) ? 15u : 3u * (2u - ((best >> 2) & 3u)) + (2u - (best & 3u));
          continue;
        }
        float m = pv[0][j];
        int r = pr[0][j], col = 0;
        // a later column wins only with a larger value, or an equal one in an EARLIER row (row-major scan order)
        bool t1 = pv[1][j] > m || (pv[1][j] == m && pr[1][j] < r);
        m = t1 ? pv[1][j] : m;
        r = t1 ? pr[1][j] : r;
        col = t1 ? 1 : col;
        bool t2 = cv[j] > m || (cv[j] == m && cr[j] < r);
        m = t2 ? cv[j] : m;
        r = t2 ? cr[j] : r;
        col = t2 ? 2 : col;
        o.set(j, m);
        am[j] = !(m > 0.f) ? 15u : (unsigned)(3 * r + col);
      }
      *reinterpret_cast<decltype(o.raw)*>(yrow + (long long)wo * C) = o.raw;
      if (irow != nullptr) {
        unsigned char* dst = irow + (long long)wo * C;
        const unsigned q0 = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
        if constexpr (CH == 8) {
          const unsigned q1 = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
          *reinterpret_cast<uint2*>(dst) = make_uint2(q0, q1);
        } else {
          *reinterpret_cast<unsigned*>(dst) = q0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      pv[0][j] = pv[1][j];
      pr[0][j] = pr[1][j];
      pv[1][j] = cv[j];
      pr[1][j] = cr[j];
      pk[0][j] = pk[1][j];
      pk[1][j] = ck[j];
    }
  }
}

// ------------------------------------------------------------------ BN coefficients
// stage 1: rows -> S partial rows;  grid (ceil(C/64), S), block 256 = 64 channels x 4 row lanes
__global__ __launch_bounds__(NT) void bn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int rows,
                                                       int C2) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int S = gridDim.y;
  const int per = (rows + S - 1) / S;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float a = 0.f;
  if (c < C2)
    for (int r = r0 + rl; r < r1; r += 4) a += partial[(long long)r * C2 + c];
  red[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && c < C2) out[(long long)blockIdx.y * C2 + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one block per 64 channels: 1024 threads = 64 channels x 16 row lanes (coalesced partial rows, four loads in flight per
// lane), LDS combine in a fixed order.  (With 4 row lanes the 256 partial rows of a layer-3 BatchNorm were a chain of 64
// dependent loads per thread: 24 us for a 1.5 MB reduction, five times per step.)
constexpr int FIN_RL = 16;
__global__ __launch_bounds__(64 * FIN_RL) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, float count,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, float momentum, float* running_mean, float* running_var,
                                                                 long long* num_batches_tracked, float* scale, float* shift,
                                                                 float* save_mean, float* save_rstd) {
  __shared__ double red[2][FIN_RL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int r = rl;
    for (; r + 3 * FIN_RL < rows; r += 4 * FIN_RL) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[(long long)(r + u * FIN_RL) * 2 * C + c];
        b[u] = partial[(long long)(r + u * FIN_RL) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1 += a[u];
        s2 += b[u];
      }
    }
    for (; r < rows; r += FIN_RL) {
      s1 += partial[(long long)r * 2 * C + c];
      s2 += partial[(long long)r * 2 * C + C + c];
    }
  }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  if (rl != 0 || c >= C) return;
  s1 = s2 = 0.0;
#pragma unroll
  for (int k = 0; k < FIN_RL; ++k) {
    s1 += red[0][k][cl];
    s2 += red[1][k][cl];
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_rstd) save_rstd[c] = rstd;
  if (running_mean) {
    const double unb = count > 1.f ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, float* rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  if (rstd != nullptr) rstd[c] = 1.0f / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// ------------------------------------------------------------------ BN apply (+residual, +ReLU)
// The per-channel coefficients sit in LDS and the channel index of a thread's vector advances by a constant per grid
// stride (per element loads from global and a 64-bit modulo per vector were most of the kernel's instructions).
// MASK (bfloat16, 8 elements per vector): byte i of `mask` = the signs of vector i's outputs, bit j set where y[8 i + j] > 0
// -- the ReLU's backward reads this bit instead of y (HtrvtGemmDesc.relu_bits).
// STREAM: inputs by non-temporal loads, for activations of >= bn_stream_bytes() (common.h)
template <typename T, int RES, bool MASK = false, bool STREAM = false>  // RES: 0 none, 1 identity residual, 2 residual with its own BN coefficients
__global__ __launch_bounds__(NT) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const T* __restrict__ res,
                                                      const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                      T* __restrict__ y, long long nvec, int C, int relu,
                                                      unsigned char* __restrict__ mask = nullptr) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sco[];   // [2 or 4][C]: scale, shift [, rscale, rshift]
  for (int k = threadIdx.x; k < C; k += NT) {
    sco[k] = scale[k];
    sco[C + k] = shift[k];
    if constexpr (RES == 2) {
      sco[2 * C + k] = rscale[k];
      sco[3 * C + k] = rshift[k];
    }
  }
  __syncthreads();
  const int cvec = C / CH;
  const long long stride = (long long)gridDim.x * NT;
  const int step = (int)(stride % cvec);
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  int cv = (int)(i % cvec);
  for (; i < nvec; i += stride) {
    const float* ca = sco + cv * CH;
    Vec16<T> v, r, o;
    if constexpr (STREAM) {
      v.raw = ld_stream16(reinterpret_cast<const decltype(v.raw)*>(x) + i);
      if constexpr (RES != 0) r.raw = ld_stream16(reinterpret_cast<const decltype(r.raw)*>(res) + i);
    } else {
      v.raw = reinterpret_cast<const decltype(v.raw)*>(x)[i];
      if constexpr (RES != 0) r.raw = reinterpret_cast<const decltype(r.raw)*>(res)[i];
    }
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float a = fmaf(v.get(j), ca[j], ca[C + j]);
      if constexpr (RES == 1) a += r.get(j);
      if constexpr (RES == 2) a += fmaf(r.get(j), ca[2 * C + j], ca[3 * C + j]);
      if (relu) a = fmaxf(a, 0.f);
      o.set(j, a);
      if constexpr (MASK) bits |= (o.get(j) > 0.f ? 1u : 0u) << j;     // of the ROUNDED output: what a reader of y would see
    }
    reinterpret_cast<decltype(o.raw)*>(y)[i] = o.raw;
    if constexpr (MASK) mask[i] = (unsigned char)bits;
    cv += step;
    if (cv >= cvec) cv -= cvec;
  }
}

// ------------------------------------------------------------------ BN + ReLU + maxpool 3x3 s(2,1) p1
// One thread per (image, column, 16-byte channel vector) MARCHES down the rows: the horizontal first-maximum of
// each input row is computed once (it serves the two vertically overlapping windows), BN + ReLU is applied once
// per loaded element, and every step issues its 6 loads up front.  (A thread-per-output version spent ~1160
// instructions per vector and was VALU-issue bound at 1.7 TB/s.)  First maximum in (row, column) scan order, as
// ATen's max_pool2d; selects only, no branches.
template <typename T>
__global__ __launch_bounds__(NT) void bn_relu_maxpool_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y,
                                                             unsigned char* __restrict__ idx, int B, int H, int W,
                                                             int C) {
  constexpr int CH = Vec16<T>::N;
  using Raw = decltype(Vec16<T>().raw);
  const int cvec = C / CH, Ho = (H - 1) / 2 + 1;
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= B * W * cvec) return;
  const int cv = i % cvec, t = i / cvec;
  const int wo = t % W, b = t / W;
  const bool okL = wo > 0, okR = wo < W - 1;
  const int wl = max(wo - 1, 0), wr = min(wo + 1, W - 1);
  float sc[CH], sf[CH], pm[CH];
  unsigned pi[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    sc[j] = scale ? scale[cv * CH + j] : 1.f;
    sf[j] = scale ? shift[cv * CH + j] : 0.f;
    pm[j] = -INFINITY;  // row -1 is padding
    pi[j] = 0;
  }
  const Raw* xb = reinterpret_cast<const Raw*>(x) + (long long)b * H * W * cvec + cv;
  Raw* yb = reinterpret_cast<Raw*>(y) + ((long long)b * Ho * W + wo) * cvec + cv;
  unsigned char* ib = idx ? idx + (((long long)b * Ho * W + wo) * cvec + cv) * CH : nullptr;

  auto row_max = [&](const Vec16<T>& l, const Vec16<T>& c, const Vec16<T>& r, bool rowok, float(&hm)[CH], unsigned(&hi)[CH]) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float v0 = fmaf(l.get(j), sc[j], sf[j]), v1 = fmaf(c.get(j), sc[j], sf[j]), v2 = fmaf(r.get(j), sc[j], sf[j]);
      if (scale) {
        v0 = fmaxf(v0, 0.f);
        v1 = fmaxf(v1, 0.f);
        v2 = fmaxf(v2, 0.f);
      }
      v0 = okL ? v0 : -INFINITY;
      v2 = okR ? v2 : -INFINITY;
      const bool t1 = v1 > v0;
      const float m01 = t1 ? v1 : v0;
      const bool t2 = v2 > m01;
      hm[j] = rowok ? (t2 ? v2 : m01) : -INFINITY;
      hi[j] = t2 ? 2u : (t1 ? 1u : 0u);
    }
  };

  for (int ho = 0; ho < Ho; ++ho) {
    const int r1 = 2 * ho, r2 = min(2 * ho + 1, H - 1);
    const bool ok2 = 2 * ho + 1 < H;
    Vec16<T> a0, a1, a2, b0, b1, b2;
    a0.raw = xb[(long long)(r1 * W + wl) * cvec];
    a1.raw = xb[(long long)(r1 * W + wo) * cvec];
    a2.raw = xb[(long long)(r1 * W + wr) * cvec];
    b0.raw = xb[(long long)(r2 * W + wl) * cvec];
    b1.raw = xb[(long long)(r2 * W + wo) * cvec];
    b2.raw = xb[(long long)(r2 * W + wr) * cvec];
    float m1[CH], m2[CH];
    unsigned i1[CH], i2[CH];
    row_max(a0, a1, a2, true, m1, i1);
    row_max(b0, b1, b2, ok2, m2, i2);
    Vec16<T> o;
    unsigned am[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float m = pm[j];
      unsigned a = pi[j];
      const bool t1 = m1[j] > m;
      m = t1 ? m1[j] : m;
      a = t1 ? 3u + i1[j] : a;
      const bool t2 = m2[j] > m;
      m = t2 ? m2[j] : m;
      a = t2 ? 6u + i2[j] : a;
      o.set(j, m);
      am[j] = (scale && !(m > 0.f)) ? 15u : a;   // 15: ReLU closed at the arg-max -> no gradient
      pm[j] = m2[j];  // input row 2*ho+1 is the top row of the next window
      pi[j] = i2[j];
    }
    yb[(long long)ho * W * cvec] = o.raw;
    if (ib != nullptr) {
      const unsigned w0 = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
      unsigned char* dst = ib + (long long)ho * W * cvec * CH;
      if constexpr (CH == 8) {
        const unsigned w1 = am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24);
        *reinterpret_cast<uint2*>(dst) = make_uint2(w0, w1);
      } else {
        *reinterpret_cast<unsigned*>(dst) = w0;
      }
    }
  }
}

// ------------------------------------------------------------------ final maxpool + span mask + pos-embed -> tokens
template <typename T>
__global__ __launch_bounds__(NT) void pool_tokens_kernel(const T* __restrict__ x, const float* __restrict__ keep,
                                                         const float* __restrict__ mask_token,
                                                         const float* __restrict__ pos, T* __restrict__ tok, int B, int H,
                                                         int W, int D) {
  constexpr int CH = Vec16<T>::N;
  const int cvec = D / CH, Ho = (H - 1) / 2 + 1, N = Ho * W;
  const long long total = (long long)B * N * cvec;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int cv = (int)(i % cvec);
    long long t = i / cvec;
    const int n = (int)(t % N), b = (int)(t / N);
    const int ho = n / W, wo = n - ho * W;
    float m[CH];
    const bool kept = keep == nullptr || keep[n] != 0.f;
    if (kept) {
#pragma unroll
      for (int j = 0; j < CH; ++j) m[j] = -INFINITY;
      for (int dy = -1; dy <= 1; ++dy) {
        const int hi = 2 * ho + dy;
        if (hi < 0 || hi >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int wi = wo + dx;
          if (wi < 0 || wi >= W) continue;
          Vec16<T> v;
          v.raw = reinterpret_cast<const decltype(v.raw)*>(x)[(((long long)b * H + hi) * W + wi) * cvec + cv];
#pragma unroll
          for (int j = 0; j < CH; ++j) m[j] = fmaxf(m[j], v.get(j));
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) m[j] = mask_token[cv * CH + j];
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < CH; ++j) o.set(j, m[j] + pos[(long long)n * D + cv * CH + j]);
    reinterpret_cast<decltype(o.raw)*>(tok)[i] = o.raw;
  }
}

inline int grid_for(long long work_items) {
  long long g = (work_items + NT - 1) / NT;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int htrvt_img_stats(const void* img, float* stats, int B, int HW, float eps, int img_u8, void* stream) {
  HTRVT_REQUIRE(HW % 4 == 0 && B > 0, "htrvt_img_stats: HW must be a multiple of 4");
  hipLaunchKernelGGL(img_stats_kernel, dim3(B), dim3(NT_IMG), 0, (hipStream_t)stream, img, stats, HW, eps, img_u8);
  return check_launch("img_stats");
}

extern "C" int htrvt_conv1_fwd(const void* img, const float* stats, const float* w, void* out, float* colstats, int B,
                               int H, int W, int C, int dtype, int img_u8, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0 && C / ch <= NT && H % 2 == 0, "htrvt_conv1_fwd: C=%d must be a multiple of %d and <= %d", C, ch,
                NT * ch);
  const int lanes = C / ch, nthr = (NT / lanes) * lanes;
  size_t smem = (size_t)3 * (W + 2) * 4;
  const size_t red = (size_t)nthr * 2 * ch * 4;
  if (red > smem) smem = red;
  HTRVT_REQUIRE(smem <= 64 * 1024, "htrvt_conv1_fwd: W=%d too wide for the LDS row buffer", W);
  dim3 grid(B * (H / 2));
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(conv1_fwd_kernel<bf16_t>, grid, dim3(NT), smem, (hipStream_t)stream, img, stats, w, (bf16_t*)out,
                       colstats, H, W, C, nthr, img_u8);
  else
    hipLaunchKernelGGL(conv1_fwd_kernel<float>, grid, dim3(NT), smem, (hipStream_t)stream, img, stats, w, (float*)out,
                       colstats, H, W, C, nthr, img_u8);
  return check_launch("conv1_fwd");
}

extern "C" int htrvt_stem_stats_rows(int B, int H) { return B * ((H / 2 + MOM_R - 1) / MOM_R); }

// (sum, sum of squares) of the conv1 output per channel, from the image alone -> colstats [2][C] (one partial row for
// htrvt_bn_finalize); partial: float32 [htrvt_stem_stats_rows(B, H)][64] workspace
extern "C" int htrvt_stem_stats(const void* img, const float* stats, const float* w, float* partial, float* colstats, int B,
                                int H, int W, int C, int img_u8, void* stream) {
  HTRVT_REQUIRE(img && stats && w && partial && colstats, "htrvt_stem_stats: null argument");
  HTRVT_REQUIRE(B > 0 && H >= 2 && H % 2 == 0 && W > 0 && C > 0, "htrvt_stem_stats: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
  const size_t smem = (size_t)(2 * MOM_R + 1) * (W + 2) * 4;
  HTRVT_REQUIRE(smem <= 160 * 1024, "htrvt_stem_stats: W=%d too wide for the LDS row buffer", W);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_moments_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int nrows = htrvt_stem_stats_rows(B, H);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stem_moments_kernel, dim3(nrows), dim3(NT), smem, st, img, stats, partial, H, W, img_u8);
  hipLaunchKernelGGL(stem_stats_kernel, dim3(1), dim3(64 * SS_RL), 0, st, partial, nrows, w, colstats, C);
  return check_launch("stem_stats");
}

// conv1 -> BatchNorm (scale / shift) -> ReLU -> max-pool in one pass over the image: y [B][Hp][W][C], idx (or NULL)
extern "C" int htrvt_stem_fwd(const void* img, const float* stats, const float* w, const float* scale, const float* shift,
                              void* y, uint8_t* idx, int B, int H, int W, int C, int dtype, int img_u8, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(img && stats && w && scale && shift && y, "htrvt_stem_fwd: null argument");
  HTRVT_REQUIRE(C % ch == 0 && C / ch <= NT && H % 2 == 0 && H >= 4, "htrvt_stem_fwd: C=%d must be a multiple of %d and <= %d", C,
                ch, NT * ch);
  if (dtype == HTRVT_BF16 && !stem_force_valu()) {   // conv1 as an MFMA product, pooling in the accumulator layout (stem_mfma.hip)
    const int r = stem_mfma_try_launch(img, stats, w, scale, shift, y, idx, B, H, W, C, img_u8, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  const int lanes = C / ch, nthr = (NT / lanes) * lanes;
  const size_t smem = (size_t)7 * (W + 2) * 4;
  HTRVT_REQUIRE(smem <= 160 * 1024, "htrvt_stem_fwd: W=%d too wide for the LDS row buffer", W);
  if (smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1;
  dim3 grid(B * Hp);
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(stem_fused_fwd_kernel<bf16_t>, grid, dim3(NT), smem, (hipStream_t)stream, img, stats, w, scale, shift,
                       (bf16_t*)y, idx, H, W, C, nthr, img_u8);
  else
    hipLaunchKernelGGL(stem_fused_fwd_kernel<float>, grid, dim3(NT), smem, (hipStream_t)stream, img, stats, w, scale, shift,
                       (float*)y, idx, H, W, C, nthr, img_u8);
  return check_launch("stem_fwd");
}

extern "C" int htrvt_bn_finalize(const float* partial, int rows, int C, float count, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                                 float* save_rstd, void* stream) {
  // caller guarantees 64 scratch rows after `rows` rows of `partial` when rows > 256
  const float* src = partial;
  int r = rows;
  if (rows > 256) {
    float* scratch = const_cast<float*>(partial) + (long long)rows * 2 * C;
    dim3 grid((2 * C + 63) / 64, 64);
    hipLaunchKernelGGL(bn_reduce_kernel, grid, dim3(NT), 0, (hipStream_t)stream, partial, scratch, rows, 2 * C);
    src = scratch;
    r = 64;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64 * FIN_RL), 0, (hipStream_t)stream, src, r, C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, (long long*)num_batches_tracked, scale, shift, save_mean, save_rstd);
  return check_launch("bn_finalize");
}

extern "C" int htrvt_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* scale, float* shift, float* rstd, int C,
                                    void* stream) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, scale, shift, rstd, C);
  return check_launch("bn_eval_coeffs");
}

extern "C" int htrvt_bn_apply(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                              const float* rshift, void* y, int64_t npix, int C, int relu, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0, "htrvt_bn_apply: C=%d must be a multiple of %d", C, ch);
  const long long nvec = npix * (C / ch);
  const int mode = res == nullptr ? 0 : (rscale == nullptr ? 1 : 2);
  dim3 grid(grid_for(nvec));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BN_APPLY(T, R)                                                                                            \
  do {                                                                                                                   \
  if (nvec * 16 >= bn_stream_bytes())                                                                                    \
    hipLaunchKernelGGL((bn_apply_kernel<T, R, false, true>), grid, dim3(NT), (size_t)(R == 2 ? 4 : 2) * C * sizeof(float), st, (const T*)x, scale, shift, (const T*)res, rscale, \
                       rshift, (T*)y, nvec, C, relu);                                                                    \
  else                                                                                                                   \
    hipLaunchKernelGGL((bn_apply_kernel<T, R>), grid, dim3(NT), (size_t)(R == 2 ? 4 : 2) * C * sizeof(float), st, (const T*)x, scale, shift, (const T*)res, rscale, \
                       rshift, (T*)y, nvec, C, relu);                                                                    \
  } while (0)
  if (dtype == HTRVT_BF16) {
    if (mode == 0) LAUNCH_BN_APPLY(bf16_t, 0);
    else if (mode == 1) LAUNCH_BN_APPLY(bf16_t, 1);
    else LAUNCH_BN_APPLY(bf16_t, 2);
  } else {
    if (mode == 0) LAUNCH_BN_APPLY(float, 0);
    else if (mode == 1) LAUNCH_BN_APPLY(float, 1);
    else LAUNCH_BN_APPLY(float, 2);
  }
#undef LAUNCH_BN_APPLY
  return check_launch("bn_apply");
}

extern "C" int htrvt_bn_apply_mask(const void* x, const float* scale, const float* shift, const void* res, const float* rscale,
                                   const float* rshift, void* y, uint8_t* mask, int64_t npix, int C, int relu, int dtype,
                                   void* stream) {
  HTRVT_REQUIRE(dtype == HTRVT_BF16 && mask != nullptr, "htrvt_bn_apply_mask: bfloat16 tensors and a mask buffer");
  HTRVT_REQUIRE(C % 8 == 0, "htrvt_bn_apply_mask: C=%d must be a multiple of 8", C);
  const long long nvec = npix * (C / 8);
  const int mode = res == nullptr ? 0 : (rscale == nullptr ? 1 : 2);
  dim3 grid(grid_for(nvec));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BN_APPLY_MASK(R)                                                                                              \
  do {                                                                                                                       \
  if (nvec * 16 >= bn_stream_bytes())                                                                                        \
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t, R, true, true>), grid, dim3(NT), (size_t)(R == 2 ? 4 : 2) * C * sizeof(float), st, \
                       (const bf16_t*)x, scale, shift, (const bf16_t*)res, rscale, rshift, (bf16_t*)y, nvec, C, relu, mask);  \
  else                                                                                                                       \
    hipLaunchKernelGGL((bn_apply_kernel<bf16_t, R, true>), grid, dim3(NT), (size_t)(R == 2 ? 4 : 2) * C * sizeof(float), st,  \
                       (const bf16_t*)x, scale, shift, (const bf16_t*)res, rscale, rshift, (bf16_t*)y, nvec, C, relu, mask); \
  } while (0)
  if (mode == 0) LAUNCH_BN_APPLY_MASK(0);
  else if (mode == 1) LAUNCH_BN_APPLY_MASK(1);
  else LAUNCH_BN_APPLY_MASK(2);
#undef LAUNCH_BN_APPLY_MASK
  return check_launch("bn_apply_mask");
}

extern "C" int htrvt_bn_relu_maxpool(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, int B,
                                     int H, int W, int C, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0, "htrvt_bn_relu_maxpool: C=%d must be a multiple of %d", C, ch);
  HTRVT_REQUIRE((long long)B * W * (C / ch) < (1ll << 31), "htrvt_bn_relu_maxpool: too many columns");
  dim3 grid((unsigned)(((long long)B * W * (C / ch) + NT - 1) / NT));
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, scale,
                       shift, (bf16_t*)y, idx, B, H, W, C);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, scale, shift,
                       (float*)y, idx, B, H, W, C);
  return check_launch("bn_relu_maxpool");
}

extern "C" int htrvt_pool_tokens(const void* x, const float* keep, const float* mask_token, const float* pos, void* tok,
                                 int B, int H, int N, int D, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  const int Ho = (H - 1) / 2 + 1;
  HTRVT_REQUIRE(D % ch == 0 && N % Ho == 0, "htrvt_pool_tokens: bad shape D=%d N=%d H=%d", D, N, H);
  const int W = N / Ho;
  const long long total = (long long)B * N * (D / ch);
  dim3 grid(grid_for(total));
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(pool_tokens_kernel<bf16_t>, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, keep,
                       mask_token, pos, (bf16_t*)tok, B, H, W, D);
  else
    hipLaunchKernelGGL(pool_tokens_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)x, keep, mask_token,
                       pos, (float*)tok, B, H, W, D);
  return check_launch("pool_tokens");
}
