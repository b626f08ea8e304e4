// conv1_bwd.hip -- backward of the stem's first stage, conv1 (Cin = 1) -> BatchNorm (train) -> ReLU ->
// max_pool2d(3, stride (2,1), pad 1) (reference resnet18.py:74-77 under autograd), with respect to conv1.weight,
// bn1.weight and bn1.bias, in ONE pass over the pooled gradient.
//
// Because conv1 has a single input channel, everything the chain rule needs is a handful of per-channel sums:
//   g      = dpool at a pooled element whose arg-max position `pos` has a positive activation, else 0
//   S[c]   = sum g                      G[c][t] = sum g * xw_t(pos)        (xw_t = whitened image tap t at pos)
//   X[t]   = sum_pos xw_t(pos)          R[t][u] = sum_pos xw_t(pos) xw_u(pos)   (image only, all conv positions)
// and then, with y = conv output, yhat = (y - mean) rstd, M = number of conv positions per channel:
//   sum g*y   = sum_t W[c][t] G[c][t]          dbeta = S          dgamma = Q = rstd (sum g*y - mean S)
//   dW[c][t]  = gamma rstd ( G[c][t] - S/M X[t] - Q/M rstd ( sum_u W[c][u] R[u][t] - mean X[t] ) )
// which replaces: scatter through the arg-max (1.6 GB written), the BatchNorm-backward reduction and apply passes
// over the conv1-sized tensor, and the conv1 weight-gradient pass over it.
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NIMG = 54;  // 9 tap sums + 45 upper-triangle tap products

// One block per pooled row (b, ph).  LDS holds the 7 whitened image rows the 3 conv rows of this pooling window row
// touch (zero halo = zero padding in whitened space).  A wave is CGW channel vectors x 64/CGW pixel lanes: adjacent
// lanes read adjacent 16-byte channel vectors of one pixel (coalesced), the pixel lanes march along the row.
template <typename T>
__global__ __launch_bounds__(512) void conv1_bwd_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                         const T* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                         float* __restrict__ partial, int H, int W, int C, int cgw,
                                                         int ld, int u8) {
  constexpr int CH = Vec16<T>::N;
  using Raw = decltype(Vec16<T>().raw);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rows = reinterpret_cast<float*>(smem_raw);  // [7][W+4]
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1, WP = W + 4;
  float* red = rows + 7 * WP;                        // [8][NIMG]
  const int b = blockIdx.x / Hp, ph = blockIdx.x - b * Hp;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  for (int i = threadIdx.x; i < 7 * WP; i += blockDim.x) {
    const int r = i / WP, c = i - r * WP;
    const int hi = 4 * ph - 3 + r, wi = c - 2;
    float v = 0.f;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (load_pixel(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
    rows[i] = v;
  }
  __syncthreads();
  const int cvec = C / CH;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cgl = lane & (cgw - 1), pl = lane / cgw, PL = 64 / cgw;
  const int cg = wv * cgw + cgl;
  float acc[CH][10];
#pragma unroll
  for (int j = 0; j < CH; ++j)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[j][t] = 0.f;
  {
    const Raw* db = reinterpret_cast<const Raw*>(dpool) + (long long)blockIdx.x * W * cvec + cg;
    const unsigned char* ib = idx + ((long long)blockIdx.x * W * cvec + cg) * CH;
    for (int px = pl; px < W; px += PL) {
      Vec16<T> vd;
      vd.raw = db[(long long)px * cvec];
      unsigned wd[2];
      if constexpr (CH == 8) {
        const uint2 u = *reinterpret_cast<const uint2*>(ib + (long long)px * cvec * CH);
        wd[0] = u.x;
        wd[1] = u.y;
      } else {
        wd[0] = *reinterpret_cast<const unsigned*>(ib + (long long)px * cvec * CH);
        wd[1] = 0;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        unsigned a = (wd[j >> 2] >> (8 * (j & 3))) & 0xffu;   // 3*row + col of the arg-max, 15 = no gradient
        const bool ok = a < 9u;
        const float g = ok ? vd.get(j) : 0.f;
        a = ok ? a : 0u;
        const unsigned i3 = (a * 11u) >> 5, j3 = a - 3u * i3;
        // conv position (2ph-1+i3, px-1+j3); tap (dh,dw) reads image (4ph-3+2*i3+dh, px-2+j3+dw) = rows[2*i3+dh][px+j3+dw]
        const float* p = rows + (2 * i3) * WP + j3 + px;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) acc[j][dh * 3 + dw] = fmaf(g, p[dh * WP + dw], acc[j][dh * 3 + dw]);
        acc[j][9] += g;
      }
    }
  }
  for (int s = cgw; s < 64; s <<= 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int t = 0; t < 10; ++t) acc[j][t] += __shfl_xor(acc[j][t], s, 64);
  }
  float* prow = partial + (long long)blockIdx.x * ld;
  if (pl == 0) {
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
      for (int t = 0; t < 10; ++t) prow[(cg * CH + j) * 10 + t] = acc[j][t];
  }

  // image-only sums over the two conv rows this block owns (h = 2ph, 2ph+1  <->  window rows i = 1, 2)
  float xs[NIMG];
#pragma unroll
  for (int k = 0; k < NIMG; ++k) xs[k] = 0.f;
  for (int q = threadIdx.x; q < 2 * W; q += blockDim.x) {
    const int hi_row = q >= W ? 1 : 0;
    const int w = q - hi_row * W;
    if (2 * ph + hi_row >= Hc) continue;
    const float* p = rows + (2 * (1 + hi_row)) * WP + w + 1;
    float xin[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) xin[dh * 3 + dw] = p[dh * WP + dw];
    int k = 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      xs[t] += xin[t];
#pragma unroll
      for (int u = t; u < 9; ++u, ++k) xs[k] = fmaf(xin[t], xin[u], xs[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NIMG; ++k) xs[k] = wave_sum(xs[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NIMG; ++k) red[wv * NIMG + k] = xs[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < NIMG) {
    float a = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) a += red[w * NIMG + threadIdx.x];
    prow[C * 10 + threadIdx.x] = a;
  }
}

// rows -> S partial rows;  grid (ceil(ld/64), S), block 256 = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void conv1_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int nrows, int ld) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int S = gridDim.y;
  const int per = (nrows + S - 1) / S;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  float a = 0.f;
  if (c < ld)
    for (int r = r0 + rl; r < r1; r += 4) a += partial[(long long)r * ld + c];
  red[rl][threadIdx.x & 63] = a;
  __syncthreads();
  if (rl == 0 && c < ld)
    out[(long long)blockIdx.y * ld + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(512) void conv1_bwd_finalize_kernel(const float* __restrict__ part, int S, int ld, int C,
                                                                 double count, const float* __restrict__ w,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* dw, float* dgamma,
                                                                 float* dbeta) {
  // all S partial rows are summed first, every thread owning whole columns (coalesced loads, independent chains),
  // into LDS; the per-channel algebra then reads its ten sums from there.  (One thread per channel walking S strided
  // rows ten times was a chain of 320 dependent loads: 53 us.)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* tot = reinterpret_cast<double*>(smem_raw);   // [ld]
  __shared__ double X[9], R[9][9];
  for (int i = threadIdx.x; i < ld; i += blockDim.x) {
    double a = 0.0;
    int s = 0;
    for (; s + 8 <= S; s += 8) {   // eight loads in flight, summed in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long long)(s + u) * ld + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < S; ++s) a += part[(long long)s * ld + i];
    tot[i] = a;
  }
  __syncthreads();
  if (threadIdx.x < NIMG) {
    const double a = tot[C * 10 + threadIdx.x];
    if (threadIdx.x < 9) {
      X[threadIdx.x] = a;
    } else {
      int k = threadIdx.x - 9, t = 0;
      while (k >= 9 - t) {
        k -= 9 - t;
        ++t;
      }
      R[t][t + k] = a;
      R[t + k][t] = a;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double G[10];
    for (int t = 0; t < 10; ++t) G[t] = tot[c * 10 + t];
    const double Ssum = G[9], mu = mean[c], r = rstd[c], gam = gamma[c];
    double gy = 0.0;
    for (int t = 0; t < 9; ++t) gy += (double)w[c * 9 + t] * G[t];
    const double Q = r * (gy - mu * Ssum);
    dbeta[c] += (float)Ssum;
    dgamma[c] += (float)Q;
    for (int t = 0; t < 9; ++t) {
      double wr = 0.0;
      for (int u = 0; u < 9; ++u) wr += (double)w[c * 9 + u] * R[u][t];
      dw[c * 9 + t] += (float)(gam * r * (G[t] - Ssum / count * X[t] - Q / count * r * (wr - mu * X[t])));
    }
  }
}

int pick_cgw(int cvec) {
  int cgw = 1;
  while (cgw < 8 && cvec % (cgw * 2) == 0) cgw *= 2;
  return cgw;
}

constexpr int S_ROWS = 32;

}  // namespace

extern "C" int htrvt_conv1_bwd_row_floats(int C) { return (C * 10 + NIMG + 63) / 64 * 64; }

extern "C" int htrvt_conv1_bwd_rows(int B, int H) {
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1;
  return B * Hp + S_ROWS;
}

extern "C" int htrvt_conv1_bwd(const void* img, const float* stats, const void* dpool, const uint8_t* idx, const float* w,
                               const float* gamma, const float* mean, const float* rstd, float* partial, float* dw,
                               float* dgamma, float* dbeta, int B, int H, int W, int C, int dtype, int img_u8, void* stream) {
  const int CH = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(dtype == HTRVT_BF16 || dtype == HTRVT_F32, "conv1_bwd: bad dtype %d", dtype);
  HTRVT_REQUIRE(img && stats && dpool && idx && w && gamma && mean && rstd && partial && dw && dgamma && dbeta,
                "conv1_bwd: null argument (train-mode batch statistics are required)");
  HTRVT_REQUIRE(B > 0 && H >= 4 && H % 2 == 0 && W > 0 && C > 0 && C % CH == 0, "conv1_bwd: bad shape B=%d H=%d W=%d C=%d", B,
                H, W, C);
  const int cvec = C / CH, cgw = pick_cgw(cvec);
  const int nw = cvec / cgw;
  HTRVT_REQUIRE(nw <= 8, "conv1_bwd: C=%d needs %d waves per block (max 8)", C, nw);
  const int Hc = H / 2, Hp = (Hc - 1) / 2 + 1;
  const int ld = htrvt_conv1_bwd_row_floats(C);
  const size_t smem = (size_t)(7 * (W + 4) + 8 * NIMG) * sizeof(float);
  HTRVT_REQUIRE(smem <= 160 * 1024, "conv1_bwd: W=%d does not fit LDS", W);
  const int nrows = B * Hp;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == HTRVT_BF16) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem);
    conv1_bwd_kernel<bf16_t><<<nrows, nw * 64, smem, st>>>(img, stats, (const bf16_t*)dpool, idx, partial, H, W, C, cgw, ld, img_u8);
  } else {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem);
    conv1_bwd_kernel<float><<<nrows, nw * 64, smem, st>>>(img, stats, (const float*)dpool, idx, partial, H, W, C, cgw, ld, img_u8);
  }
  float* red = partial + (long long)nrows * ld;
  conv1_bwd_reduce_kernel<<<dim3((ld + 63) / 64, S_ROWS), 256, 0, st>>>(partial, red, nrows, ld);
  HTRVT_REQUIRE((size_t)ld * sizeof(double) <= 64 * 1024, "conv1_bwd: C=%d too wide for the finalize kernel's LDS", C);
  conv1_bwd_finalize_kernel<<<1, 512, (size_t)ld * sizeof(double), st>>>(red, S_ROWS, ld, C, (double)B * Hc * W, w, gamma, mean, rstd, dw,
                                                                        dgamma, dbeta);
  return check_launch("conv1_bwd");
}
