// bwd.hip -- HBM-bound backward kernels of the hot path (autograd of reference
// HTR_VT.py:222-241 / resnet18.py:23-39,73-84, i.e. what train.py:123
// `loss.backward()` runs in ATen): LayerNorm / softmax / sequence-whiten backward,
// train-mode BatchNorm backward (reduce + apply, fused with the ReLU mask),
// max-pool backward (index based), token-assembly backward, column sums for
// bias gradients and the Cin=1 conv1 weight gradient.
// Every parameter-gradient output is float32 and is ACCUMULATED (+=).
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT = 256;
constexpr int MAXC = 4;

inline int grid_for(long long work_items, int cap = 256 * 8) {
  long long g = (work_items + NT - 1) / NT;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ sequence whiten backward
// y = (x-mu)*r over NC elements: dx = r*(dy - mean(dy) - y*mean(dy*y))
template <typename T>
__global__ __launch_bounds__(NT) void seq_whiten_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ stats, T* __restrict__ dx, int NC,
                                                            int C, int ldo) {
  __shared__ float red[8];
  const float* dys = dy + (long long)blockIdx.x * NC;
  const float* ys = y + (long long)blockIdx.x * NC;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < NC; i += NT) {
    s1 += dys[i];
    s2 += dys[i] * ys[i];
  }
  const float m1 = block_sum_256(s1, red) / (float)NC;
  const float m2 = block_sum_256(s2, red) / (float)NC;
  const float r = stats[2 * blockIdx.x + 1];
  T* dxs = dx + (long long)blockIdx.x * (NC / C) * ldo;
  for (int i = threadIdx.x; i < NC; i += NT) {
    const int n = i / C, c = i - n * C;
    dxs[(long long)n * ldo + c] = from_f32<T>(r * (dys[i] - m1 - ys[i] * m2));
  }
}

// ------------------------------------------------------------------ LayerNorm backward
// NK = ceil(D / CH / 64): 16-byte chunks per lane, a compile-time constant so that every load of a row is in straight-line code
template <typename T, int NK>
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const T* __restrict__ dres,
                                                           T* __restrict__ dx, float* __restrict__ partial, long long rows,
                                                           int D) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [4][2*D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D / CH;
  float ag[NK][CH], ab[NK][CH];
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int j = 0; j < CH; ++j) ag[k][j] = ab[k][j] = 0.f;

  // gamma of this lane's chunks lives in registers; all loads of a row (x, dy, dres) go out together from clamped,
  // always-valid offsets (behind `if (c < nchunk)` hipcc waits for each load before issuing the next one)
  using Raw = decltype(Vec16<T>().raw);
  float gm[NK][CH];
  bool act[NK];
  int cc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c = lane + 64 * k;
    act[k] = c < nchunk;
    cc[k] = act[k] ? c : 0;
#pragma unroll
    for (int j = 0; j < CH; ++j) gm[k][j] = act[k] ? gamma[c * CH + j] : 0.f;
  }
  for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
    const float mu = mean[row], r = rstd[row];
    Vec16<T> vx[NK], vd[NK], dr[NK];
    const Raw* xr = reinterpret_cast<const Raw*>(x + row * D);
    const Raw* dyr = reinterpret_cast<const Raw*>(dy + row * D);
    const Raw* drr = reinterpret_cast<const Raw*>(dres + row * D);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      vx[k].raw = xr[cc[k]];
      vd[k].raw = dyr[cc[k]];
      if (dres) dr[k].raw = drr[cc[k]];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float xh = (vx[k].get(j) - mu) * r;
          const float d = act[k] ? vd[k].get(j) : 0.f;
          const float g = d * gm[k][j];
          s1 += g;
          s2 += g * xh;
          ag[k][j] += d * xh;
          ab[k][j] += d;
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      {
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float xh = (vx[k].get(j) - mu) * r;
          float v = r * (vd[k].get(j) * gm[k][j] - m1 - xh * m2);
          if (dres) v += dr[k].get(j);
          o.set(j, v);
        }
        if (act[k]) reinterpret_cast<Raw*>(dx + row * D)[lane + 64 * k] = o.raw;
      }
    }
  }
  // block partial of dgamma / dbeta
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[wave * 2 * D + c * CH + j] = ag[k][j];
        red[wave * 2 * D + D + c * CH + j] = ab[k][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += NT)
    partial[(long long)blockIdx.x * 2 * D + i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
}

// ------------------------------------------------------------------ softmax backward: dS = scale * P * (dP - sum(dP*P))
template <typename T>
__global__ __launch_bounds__(NT) void softmax_bwd_rows_kernel(const T* __restrict__ p, const float* __restrict__ dp,
                                                              T* __restrict__ ds, long long rows, int n, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = n / 4;
  float4 vp[MAXC], vd[MAXC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      if constexpr (sizeof(T) == 4) {
        vp[k] = reinterpret_cast<const float4*>(p + row * n)[c];
      } else {
        const uint2 u = reinterpret_cast<const uint2*>(p + row * n)[c];
        vp[k] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u));
      }
      vd[k] = reinterpret_cast<const float4*>(dp + row * n)[c];
      s += (vp[k].x * vd[k].x + vp[k].y * vd[k].y) + (vp[k].z * vd[k].z + vp[k].w * vd[k].w);
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < nchunk) {
      const float a = scale * vp[k].x * (vd[k].x - s), b = scale * vp[k].y * (vd[k].y - s);
      const float cc = scale * vp[k].z * (vd[k].z - s), d = scale * vp[k].w * (vd[k].w - s);
      if constexpr (sizeof(T) == 4) {
        reinterpret_cast<float4*>(ds + row * n)[c] = make_float4(a, b, cc, d);
      } else {
        uint2 o;
        o.x = pack_bf16x2(a, b);
        o.y = pack_bf16x2(cc, d);
        reinterpret_cast<uint2*>(ds + row * n)[c] = o;
      }
    }
  }
}

// ------------------------------------------------------------------ column sums: out[c] += sum_r x[r*ld + c]
// optional row filter: rows whose keep[(r % keep_mod)] != 0 are skipped (masked-token gradient)
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* __restrict__ x, long long rows, int cols, long long ld,
                                                      float* __restrict__ out, const float* __restrict__ keep,
                                                      int keep_mod, float* __restrict__ ws) {
  // block = 32 column chunks (16 B each) x 32 row lanes; a half-wave reads 512 contiguous bytes of one row and every
  // thread keeps 4 row loads in flight.  Reproducible: with several row splits (gridDim.y > 1) every split stores its
  // own row of partial sums in ws[split][cols]; rowsum_f32_kernel then adds the rows to `out` in split order.
  constexpr int CH = Vec16<T>::N, RL = 32;
  using Raw = decltype(Vec16<T>().raw);
  __shared__ float red[RL][32][CH];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int chunk = blockIdx.x * 32 + cl;
  const int nchunk = cols / CH;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float a[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) a[j] = 0.f;
  if (chunk < nchunk) {
    const T* xc = x + (long long)chunk * CH;
    for (long long r = r0 + rl; r < r1; r += 4 * RL) {
      Vec16<T> v[4];
      bool use[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long ru = r + RL * u;
        use[u] = ru < r1 && (keep == nullptr || keep[(unsigned)ru % (unsigned)keep_mod] == 0.f);
        v[u].raw = *reinterpret_cast<const Raw*>(xc + (use[u] ? ru : r0) * ld);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < CH; ++j) a[j] += use[u] ? v[u].get(j) : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) red[rl][cl][j] = a[j];
  __syncthreads();
  // 32 chunks x CH columns = 32*CH outputs, one thread each
  if ((int)threadIdx.x < 32 * CH) {
    const int c = threadIdx.x / CH, j = threadIdx.x - c * CH;
    if (blockIdx.x * 32 + c < nchunk) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < RL; ++k) t += red[k][c][j];
      const int col = (blockIdx.x * 32 + c) * CH + j;
      if (gridDim.y > 1)
        ws[(long long)blockIdx.y * cols + col] = t;
      else
        out[col] += t;
    }
  }
}

// ------------------------------------------------------------------ BatchNorm backward (train mode)
// pass A: partial[blk][2][C] = { sum g, sum g*xhat },  g = dy * (yact > 0 if yact)
template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ yact,
                                                           const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ partial,
                                                           long long npix, int C, int nthr) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [nthr][2*CH]
  const int cvec = C / CH, ppb = nthr / cvec;
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
  if ((int)threadIdx.x < nthr) {
    const int cv = threadIdx.x % cvec, pl = threadIdx.x / cvec;
    float mu[CH], r[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      mu[j] = mean[cv * CH + j];
      r[j] = rstd[cv * CH + j];
    }
    for (long long p = (long long)blockIdx.x * ppb + pl; p < npix; p += (long long)gridDim.x * ppb) {
      Vec16<T> vd, vx, vy;
      vd.raw = reinterpret_cast<const decltype(vd.raw)*>(dy)[p * cvec + cv];
      vx.raw = reinterpret_cast<const decltype(vx.raw)*>(x)[p * cvec + cv];
      if (yact) vy.raw = reinterpret_cast<const decltype(vy.raw)*>(yact)[p * cvec + cv];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float g = vd.get(j);
        if (yact && !(vy.get(j) > 0.f)) g = 0.f;
        s1[j] += g;
        s2[j] += g * (vx.get(j) - mu[j]) * r[j];
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      red[threadIdx.x * 2 * CH + j] = s1[j];
      red[threadIdx.x * 2 * CH + CH + j] = s2[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += NT) {
    const int cv = c / CH, j = c - cv * CH;
    float a = 0.f, q = 0.f;
    for (int k = 0; k < ppb; ++k) {
      a += red[(k * cvec + cv) * 2 * CH + j];
      q += red[(k * cvec + cv) * 2 * CH + CH + j];
    }
    partial[(long long)blockIdx.x * 2 * C + c] = a;
    partial[(long long)blockIdx.x * 2 * C + C + c] = q;
  }
}

// finalize: dgamma += sum g*xhat ; dbeta += sum g ; coefficients of dx = cA*g + cB*x + cC
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C, float count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* dgamma, float* dbeta, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < rows; ++r) {
    s1 += partial[(long long)r * 2 * C + c];
    s2 += partial[(long long)r * 2 * C + C + c];
  }
  dbeta[c] += (float)s1;
  dgamma[c] += (float)s2;
  const double g = gamma[c], r = rstd[c], mu = mean[c];
  coef[c] = (float)(g * r);
  if (count <= 0.f) {   // eval-mode BatchNorm: mean / rstd are constants (running statistics), dx = gamma * rstd * g
    coef[C + c] = 0.f;
    coef[2 * C + c] = 0.f;
    return;
  }
  coef[C + c] = (float)(-g * r * r * s2 / count);
  coef[2 * C + c] = (float)(-g * r * s1 / count + g * r * r * mu * s2 / count);
}

// pass B: dx = cA*g + cB*x + cC ; optionally also writes g (the ReLU-masked incoming gradient)
// The 3 C coefficients sit in LDS (per element loads from global were 24 of the 27 loads of a vector) and the channel
// index of a thread's vector advances by a constant per grid stride (no 64-bit modulo per vector).
// STREAM: the inputs by non-temporal loads (ld_stream16) -- activations of >= bn_stream_bytes() that no cache will hold until their
// next reader; smaller ones (layer 3: 50 MB) are served from the memory-side cache and lose with them (A/B in profiles/r05_experiments.md (m))
template <typename T, bool STREAM>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ yact,
                                                          const T* __restrict__ x, const float* __restrict__ coef,
                                                          T* __restrict__ dx, T* __restrict__ gout, long long nvec,
                                                          int C) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sco[];   // [3][C]
  for (int k = threadIdx.x; k < 3 * C; k += NT) sco[k] = coef[k];
  __syncthreads();
  const int cvec = C / CH;
  const long long stride = (long long)gridDim.x * NT;
  const int step = (int)(stride % cvec);
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  int cv = (int)(i % cvec);
  for (; i < nvec; i += stride) {
    const float* ca = sco + cv * CH;
    Vec16<T> vd, vx, vy, o, go;
    if constexpr (STREAM) {
      vd.raw = ld_stream16(reinterpret_cast<const decltype(vd.raw)*>(dy) + i);
      vx.raw = ld_stream16(reinterpret_cast<const decltype(vx.raw)*>(x) + i);
      if (yact) vy.raw = ld_stream16(reinterpret_cast<const decltype(vy.raw)*>(yact) + i);
    } else {
      vd.raw = reinterpret_cast<const decltype(vd.raw)*>(dy)[i];
      vx.raw = reinterpret_cast<const decltype(vx.raw)*>(x)[i];
      if (yact) vy.raw = reinterpret_cast<const decltype(vy.raw)*>(yact)[i];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float g = vd.get(j);
      if (yact && !(vy.get(j) > 0.f)) g = 0.f;
      go.set(j, g);
      o.set(j, fmaf(ca[j], g, fmaf(ca[C + j], vx.get(j), ca[2 * C + j])));
    }
    reinterpret_cast<decltype(o.raw)*>(dx)[i] = o.raw;
    if (gout) reinterpret_cast<decltype(go.raw)*>(gout)[i] = go.raw;
    cv += step;
    if (cv >= cvec) cv -= cvec;
  }
}

// pass B for the two BatchNorms that consume the SAME gradient (a stage's first block: bn2 of the main branch and the
// downsample BN both receive the masked gradient of the block output, resnet18.py:33-37): g is read once,
// dx1 = c1A*g + c1B*x1 + c1C and dx2 = c2A*g + c2B*x2 + c2C.  5 streams instead of 6.
template <typename T, bool STREAM>
__global__ __launch_bounds__(NT) void bn_bwd_apply2_kernel(const T* __restrict__ g, const T* __restrict__ x1, const float* __restrict__ coef1,
                                                           T* __restrict__ dx1, const T* __restrict__ x2, const float* __restrict__ coef2,
                                                           T* __restrict__ dx2, long long nvec, int C) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) float sco[];   // [2][3][C]
  for (int k = threadIdx.x; k < 3 * C; k += NT) {
    sco[k] = coef1[k];
    sco[3 * C + k] = coef2[k];
  }
  __syncthreads();
  const int cvec = C / CH;
  const long long stride = (long long)gridDim.x * NT;
  const int step = (int)(stride % cvec);
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  int cv = (int)(i % cvec);
  for (; i < nvec; i += stride) {
    const float* ca = sco + cv * CH;
    const float* cb = ca + 3 * C;
    Vec16<T> vg, v1, v2, o1, o2;
    if constexpr (STREAM) {
      vg.raw = ld_stream16(reinterpret_cast<const decltype(vg.raw)*>(g) + i);
      v1.raw = ld_stream16(reinterpret_cast<const decltype(v1.raw)*>(x1) + i);
      v2.raw = ld_stream16(reinterpret_cast<const decltype(v2.raw)*>(x2) + i);
    } else {
      vg.raw = reinterpret_cast<const decltype(vg.raw)*>(g)[i];
      v1.raw = reinterpret_cast<const decltype(v1.raw)*>(x1)[i];
      v2.raw = reinterpret_cast<const decltype(v2.raw)*>(x2)[i];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float gg = vg.get(j);
      o1.set(j, fmaf(ca[j], gg, fmaf(ca[C + j], v1.get(j), ca[2 * C + j])));
      o2.set(j, fmaf(cb[j], gg, fmaf(cb[C + j], v2.get(j), cb[2 * C + j])));
    }
    reinterpret_cast<decltype(o1.raw)*>(dx1)[i] = o1.raw;
    reinterpret_cast<decltype(o2.raw)*>(dx2)[i] = o2.raw;
    cv += step;
    if (cv >= cvec) cv -= cvec;
  }
}

// ------------------------------------------------------------------ first max-pool backward (index based) + ReLU mask
// g[b,hi,wi,c] = (x*scale+shift > 0) * sum_{windows (ho,wo) containing (hi,wi) with idx == position} dpool[b,ho,wo,c]
// One thread per (image, column, channel vector) marches down the input rows; a pooled row (gradient + arg-max
// bytes of the three windows around this column) is loaded once and serves the three input rows it covers.
template <typename T>
__global__ __launch_bounds__(NT) void maxpool_bwd_kernel(const T* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                         const T* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ g, int B, int H,
                                                         int W, int C) {
  constexpr int CH = Vec16<T>::N;
  using Raw = decltype(Vec16<T>().raw);
  const int cvec = C / CH, Ho = (H - 1) / 2 + 1;
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= B * W * cvec) return;
  const int cv = i % cvec, t = i / cvec;
  const int wi = t % W, b = t / W;
  // window column wo = wi + 1 - dx  (dx = 0, 1, 2)
  const bool okc[3] = {wi < W - 1, true, wi > 0};
  const int woc[3] = {min(wi + 1, W - 1), wi, max(wi - 1, 0)};
  float sc[CH], sf[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    sc[j] = scale[cv * CH + j];
    sf[j] = shift[cv * CH + j];
  }
  const Raw* db = reinterpret_cast<const Raw*>(dpool) + (long long)b * Ho * W * cvec + cv;
  const unsigned char* ib = idx + ((long long)b * Ho * W * cvec + cv) * CH;
  const Raw* xb = reinterpret_cast<const Raw*>(x) + ((long long)b * H * W + wi) * cvec + cv;
  Raw* gb = reinterpret_cast<Raw*>(g) + ((long long)b * H * W + wi) * cvec + cv;

  struct PRow {
    Vec16<T> d[3];
    unsigned w[3][2];
  };
  auto load_row = [&](int ho, PRow& r) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const long long o = (long long)(ho * W + woc[dx]) * cvec;
      r.d[dx].raw = db[o];
      if constexpr (CH == 8) {
        const uint2 u = *reinterpret_cast<const uint2*>(ib + o * CH);
        r.w[dx][0] = u.x;
        r.w[dx][1] = u.y;
      } else {
        r.w[dx][0] = *reinterpret_cast<const unsigned*>(ib + o * CH);
        r.w[dx][1] = 0;
      }
    }
  };
  // contribution of pooled row `r` to an input row that sits at window row `dy` of it
  auto gather = [&](const PRow& r, unsigned dy, bool rowok, float(&acc)[CH]) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const unsigned pos = dy * 3 + dx;
      const bool ok = rowok && okc[dx];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const unsigned idj = (r.w[dx][j >> 2] >> (8 * (j & 3))) & 0xffu;
        acc[j] += (ok && idj == pos) ? r.d[dx].get(j) : 0.f;
      }
    }
  };
  auto finish = [&](int hi, const float(&acc)[CH]) {
    Vec16<T> vx, o;
    vx.raw = xb[(long long)hi * W * cvec];
#pragma unroll
    for (int j = 0; j < CH; ++j) o.set(j, fmaf(vx.get(j), sc[j], sf[j]) > 0.f ? acc[j] : 0.f);
    gb[(long long)hi * W * cvec] = o.raw;
  };

  PRow cur, nxt;
  load_row(0, cur);
  for (int ho = 0; ho < Ho; ++ho) {
    const bool has_next = ho + 1 < Ho;
    load_row(has_next ? ho + 1 : ho, nxt);
    {  // input row 2*ho: centre row (dy = 1) of pooled row ho
      float acc[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] = 0.f;
      gather(cur, 1u, true, acc);
      finish(2 * ho, acc);
    }
    if (2 * ho + 1 < H) {  // input row 2*ho+1: top row (dy = 0) of pooled row ho+1, bottom row (dy = 2) of row ho
      float acc[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] = 0.f;
      gather(nxt, 0u, has_next, acc);
      gather(cur, 2u, true, acc);
      finish(2 * ho + 1, acc);
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------ token assembly backward (final max-pool)
template <typename T>
__global__ __launch_bounds__(NT) void pool_tokens_bwd_kernel(const T* __restrict__ dtok, const T* __restrict__ x,
                                                             const float* __restrict__ keep, T* __restrict__ dx, int B,
                                                             int H, int W, int D) {
  constexpr int CH = Vec16<T>::N;
  const int cvec = D / CH, Ho = (H - 1) / 2 + 1;
  const long long total = (long long)B * H * W * cvec;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int cv = (int)(i % cvec);
    long long pix = i / cvec;
    const int wi = (int)(pix % W);
    pix /= W;
    const int hi = (int)(pix % H), b = (int)(pix / H);
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < 3; ++dy) {
      const int t = hi + 1 - dy;
      if (t < 0 || (t & 1)) continue;
      const int ho = t >> 1;
      if (ho >= Ho) continue;
      for (int dx_ = 0; dx_ < 3; ++dx_) {
        const int wo = wi + 1 - dx_;
        if (wo < 0 || wo >= W) continue;
        const int n = ho * W + wo;
        if (keep != nullptr && keep[n] == 0.f) continue;
        // first maximum of window (ho,wo) in scan order
        float m[CH];
        int am[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          m[j] = -INFINITY;
          am[j] = -1;
        }
        for (int r = 0; r < 3; ++r) {
          const int h2 = 2 * ho - 1 + r;
          if (h2 < 0 || h2 >= H) continue;
          for (int c = 0; c < 3; ++c) {
            const int w2 = wo - 1 + c;
            if (w2 < 0 || w2 >= W) continue;
            Vec16<T> v;
            v.raw = reinterpret_cast<const decltype(v.raw)*>(x)[(((long long)b * H + h2) * W + w2) * cvec + cv];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float a = v.get(j);
              if (a > m[j]) {
                m[j] = a;
                am[j] = r * 3 + c;
              }
            }
          }
        }
        Vec16<T> vd;
        vd.raw = reinterpret_cast<const decltype(vd.raw)*>(dtok)[((long long)b * Ho * W + n) * cvec + cv];
#pragma unroll
        for (int j = 0; j < CH; ++j)
          if (am[j] == dy * 3 + dx_) acc[j] += vd.get(j);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < CH; ++j) o.set(j, acc[j]);
    reinterpret_cast<decltype(o.raw)*>(dx)[i] = o.raw;
  }
}

// H == 2 (every 64-pixel-high line image: one pooled row): one thread per (b, column, channel vector) loads the 2 x 5
// inputs that the three windows containing its column read, finds the three arg-maxes in registers and writes both rows'
// gradients -- 13 loads per two outputs instead of 42 (the general kernel re-derives every window per input element).
template <typename T>
__global__ __launch_bounds__(NT) void pool_tokens_bwd_h2_kernel(const T* __restrict__ dtok, const T* __restrict__ x,
                                                                const float* __restrict__ keep, T* __restrict__ dx, int B,
                                                                int W, int D) {
  constexpr int CH = Vec16<T>::N;
  using Raw = decltype(Vec16<T>().raw);
  const int cvec = D / CH;
  const long long total = (long long)B * W * cvec;
  const long long i = (long long)blockIdx.x * NT + threadIdx.x;
  if (i >= total) return;
  const int cv = (int)(i % cvec);
  const long long t = i / cvec;
  const int wi = (int)(t % W), b = (int)(t / W);
  const Raw* xr = reinterpret_cast<const Raw*>(x) + (long long)b * 2 * W * cvec + cv;
  const Raw* dr = reinterpret_cast<const Raw*>(dtok) + (long long)b * W * cvec + cv;
  // columns wi-2 .. wi+2 of both rows and the three token gradients, all in flight together (clamped addresses)
  Vec16<T> v[2][5], g[3];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int w2 = min(max(wi - 2 + c, 0), W - 1);
    v[0][c].raw = xr[(long long)w2 * cvec];
    v[1][c].raw = xr[(long long)(W + w2) * cvec];
  }
  bool live[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // window centred on column wo = wi - 1 + k
    const int wo = wi - 1 + k, woc = min(max(wo, 0), W - 1);
    live[k] = wo >= 0 && wo < W && (keep == nullptr || keep[woc] != 0.f);
    g[k].raw = dr[(long long)woc * cvec];
  }
  float acc[2][CH];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[r][j] = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // first maximum of the window in (row, column) scan order; its columns are v[.][k .. k+2], this thread's column is
    // window column 2 - k
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float m = -INFINITY;
      int am = -1;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int w2 = wi - 2 + k + c;
          const float a = (w2 >= 0 && w2 < W) ? v[r][k + c].get(j) : -INFINITY;
          if (a > m) {
            m = a;
            am = r * 3 + c;
          }
        }
      const float gv = live[k] ? g[k].get(j) : 0.f;
      if (am == 2 - k) acc[0][j] += gv;
      if (am == 3 + 2 - k) acc[1][j] += gv;
    }
  }
  Raw* dxr = reinterpret_cast<Raw*>(dx) + ((long long)b * 2 * W + wi) * cvec + cv;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < CH; ++j) o.set(j, acc[r][j]);
    dxr[(long long)r * W * cvec] = o.raw;
  }
}

// ------------------------------------------------------------------ conv1 weight gradient (Cin = 1)
// partial[blk][C*9] = sum over the block's output rows of dY[pix][c] * whitened tap
template <typename T>
__global__ __launch_bounds__(NT) void conv1_wgrad_kernel(const void* __restrict__ img, const float* __restrict__ stats,
                                                         const T* __restrict__ dy, float* __restrict__ partial, int B,
                                                         int H, int W, int C, int nthr, int u8) {
  constexpr int CH = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rows = reinterpret_cast<float*>(smem_raw);  // [3][W+2], later reduction scratch
  const int Ho = H / 2, WP = W + 2;
  const int cvec = C / CH, ppb = nthr / cvec;
  const int cg = threadIdx.x % cvec, p0 = threadIdx.x / cvec;
  float acc[CH][9];
#pragma unroll
  for (int j = 0; j < CH; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = 0.f;
  for (int row = blockIdx.x; row < B * Ho; row += gridDim.x) {
    const int b = row / Ho, ho = row - b * Ho;
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * WP; i += NT) {
      const int r = i / WP, c = i - r * WP;
      const int hi = 2 * ho - 1 + r, wi = c - 1;
      float v = 0.f;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (load_pixel(img, ((long long)b * H + hi) * W + wi, u8) - mean) * rstd;
      rows[i] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < nthr) {
      const T* drow = dy + ((long long)row * W) * C + cg * CH;
      for (int px = p0; px < W; px += ppb) {
        Vec16<T> vd;
        vd.raw = *reinterpret_cast<const decltype(vd.raw)*>(drow + (long long)px * C);
        float xin[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) xin[r * 3 + c] = rows[r * WP + px + c];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float d = vd.get(j);
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[j][t] = fmaf(d, xin[t], acc[j][t]);
        }
      }
    }
  }
  // reduce the ppb threads sharing a channel group through LDS, one tap at a time, in a fixed order (reproducible)
  float* prow = partial + (long long)blockIdx.x * C * 9;
  float* red = rows;   // [nthr][CH] floats <= 8 KB (the launcher sizes the LDS for both uses)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    if ((int)threadIdx.x < nthr) {
#pragma unroll
      for (int j = 0; j < CH; ++j) red[threadIdx.x * CH + j] = acc[j][t];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
      const int cv = c / CH, j = c - cv * CH;
      float a = 0.f;
      for (int k = 0; k < ppb; ++k) a += red[(k * cvec + cv) * CH + j];
      prow[c * 9 + t] = a;
    }
  }
}

// out[c] += sum_r partial[r][c]   (float32 rows)
// one thread per column, rows added in ascending order (reproducible); 8 independent loads in flight per thread, 64-thread
// blocks so that a few hundred columns still spread over several CUs (this runs ~40 times per step, latency-bound)
constexpr int RS_NT = 64;
__global__ __launch_bounds__(RS_NT) void rowsum_f32_kernel(const float* __restrict__ partial, int rows, int cols,
                                                           float* __restrict__ out) {
  const int c = blockIdx.x * RS_NT + threadIdx.x;
  if (c >= cols) return;
  const float* src = partial + c;
  float a = 0.f;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long long)(r + u) * cols];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; r < rows; ++r) a += src[(long long)r * cols];
  out[c] += a;
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = from_f32<T>(src[i]);
}

// decoupled weight decay Adam over one flat float32 buffer (torch.optim.AdamW semantics)
// DEV: the seven derived scalars come from device memory (`hyper`, written by the host before a captured graph is
// replayed: a graph bakes its kernel arguments, and lr / the step number change every step) instead of the argument list
template <bool DEV>
__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, float decay, float w1, float b2,
                                                   float w2, float eps, float bc2s, float neg_step,
                                                   const float* __restrict__ hyper) {
#pragma clang fp contract(off)
  if constexpr (DEV) {
    decay = hyper[0]; w1 = hyper[1]; b2 = hyper[2]; w2 = hyper[3]; eps = hyper[4]; bc2s = hyper[5]; neg_step = hyper[6];
  }
  // the statement order of torch.optim.AdamW's single-tensor step (torch/optim/adamw.py): p *= 1 - lr*wd;
  // m.lerp_(g, 1 - beta1); v = v*beta2 + (1 - beta2)*g*g; denom = sqrt(v)/sqrt(bc2) + eps; p += (-lr/bc1) * m/denom.
  // Scalars are formed in double on the host (Python floats are doubles) and rounded once.  One rounding per operation:
  // no FMA contraction in this function; `/` and sqrtf are the correctly rounded forms (hipcc default).
  // g, m, v are streamed once per step (1.5 GB with p for the 53.5 M parameters of the d768 model): non-temporal accesses,
  // so that the lines do not displace the weights the next forward re-reads through L2; p is stored normally
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = gg[j];
      float pj = pp[j] * decay;
      const float mj = mm[j] + w1 * (gj - mm[j]);
      const float vj = vv[j] * b2 + (w2 * gj) * gj;
      const float denom = sqrtf(vj) / bc2s + eps;
      pj = pj + (neg_step * mj) / denom;     // addcdiv_: self + value * t1 / t2, evaluated left to right as ATen does
      pp[j] = pj;
      mm[j] = mj;
      vv[j] = vj;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
  }
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, ...)        \
  do {                                        \
    if ((dtype) == HTRVT_BF16) {              \
      using T = bf16_t;                       \
      KERNEL;                                 \
    } else {                                  \
      using T = float;                        \
      KERNEL;                                 \
    }                                         \
  } while (0)

extern "C" int htrvt_seq_whiten_bwd(const float* dy, const float* y, const float* stats, void* dx, int B, int N, int C,
                                    int ldo, int dtype, void* stream) {
  HTRVT_REQUIRE(ldo >= C, "htrvt_seq_whiten_bwd: ldo < C");
  DISPATCH_T(dtype, hipLaunchKernelGGL(seq_whiten_bwd_kernel<T>, dim3(B), dim3(NT), 0, (hipStream_t)stream, dy, y, stats,
                                       (T*)dx, N * C, C, ldo));
  return check_launch("seq_whiten_bwd");
}

extern "C" int htrvt_layernorm_bwd_blocks(int64_t rows) {
  long long g = (rows + 3) / 4;
  if (g > 1024) g = 1024;
  return (int)g;
}

extern "C" int htrvt_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                   const void* dres, void* dx, float* partial, int64_t rows, int D, int dtype,
                                   void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(D % ch == 0 && D / ch <= 64 * MAXC, "htrvt_layernorm_bwd: D=%d unsupported", D);
  const size_t smem = (size_t)8 * D * 4;
  HTRVT_REQUIRE(smem <= 64 * 1024, "htrvt_layernorm_bwd: D=%d too large", D);
  dim3 grid(htrvt_layernorm_bwd_blocks(rows));
  const int nk = (D / ch + 63) / 64;
#define LN_BWD(NKV)                                                                                                      \
  DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T, NKV>), grid, dim3(NT), smem, (hipStream_t)stream,         \
                                       (const T*)dy, (const T*)x, mean, rstd, gamma, (const T*)dres, (T*)dx, partial,      \
                                       (long long)rows, D))
  if (nk == 1) LN_BWD(1);
  else if (nk == 2) LN_BWD(2);
  else if (nk == 3) LN_BWD(3);
  else LN_BWD(4);
#undef LN_BWD
  return check_launch("layernorm_bwd");
}

extern "C" int htrvt_softmax_bwd_rows(const void* p, const float* dp, void* ds, int64_t rows, int n, float scale, int dtype,
                                      void* stream) {
  HTRVT_REQUIRE(n % 4 == 0 && n / 4 <= 64 * MAXC, "htrvt_softmax_bwd_rows: n=%d unsupported", n);
  dim3 grid((unsigned)((rows + 3) / 4));
  DISPATCH_T(dtype, hipLaunchKernelGGL(softmax_bwd_rows_kernel<T>, grid, dim3(NT), 0, (hipStream_t)stream, (const T*)p, dp,
                                       (T*)ds, (long long)rows, n, scale));
  return check_launch("softmax_bwd_rows");
}

static int colsum_splits(int64_t rows) {
  long long splits = rows / 128;         // >= 128 rows (4 per thread) per block
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  return (int)splits;
}

extern "C" size_t htrvt_colsum_workspace_floats(int64_t rows, int cols) {
  const int s = colsum_splits(rows);
  return s > 1 ? (size_t)s * cols : 0;
}

extern "C" int htrvt_colsum(const void* x, int64_t rows, int cols, int64_t ld, float* out, const float* keep, int keep_mod,
                            int dtype, float* workspace, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(cols % ch == 0 && ld % ch == 0, "htrvt_colsum: cols/ld must be multiples of %d", ch);
  const int gx = (cols / ch + 31) / 32;
  const int splits = colsum_splits(rows);
  HTRVT_REQUIRE(splits == 1 || workspace != nullptr, "htrvt_colsum: %d rows need a workspace (htrvt_colsum_workspace_floats)",
                (int)rows);
  dim3 grid(gx, (unsigned)splits);
  DISPATCH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(1024), 0, (hipStream_t)stream, (const T*)x, (long long)rows,
                                       cols, (long long)ld, out, keep, keep_mod > 0 ? keep_mod : 1, workspace));
  if (splits > 1)
    hipLaunchKernelGGL(rowsum_f32_kernel, dim3((cols + RS_NT - 1) / RS_NT), dim3(RS_NT), 0, (hipStream_t)stream, workspace, splits, cols, out);
  return check_launch("colsum");
}

extern "C" int htrvt_rowsum_f32(const float* partial, int rows, int cols, float* out, void* stream) {
  hipLaunchKernelGGL(rowsum_f32_kernel, dim3((cols + RS_NT - 1) / RS_NT), dim3(RS_NT), 0, (hipStream_t)stream, partial, rows, cols, out);
  return check_launch("rowsum_f32");
}

extern "C" int htrvt_bn_bwd_blocks(int64_t npix) {
  long long g = npix / 64;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int htrvt_bn_bwd_reduce(const void* dy, const void* yact, const void* x, const float* mean, const float* rstd,
                                   float* partial, int64_t npix, int C, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0 && C / ch <= NT, "htrvt_bn_bwd_reduce: C=%d unsupported", C);
  HTRVT_REQUIRE(dy && x && mean && rstd && partial, "htrvt_bn_bwd_reduce: null dy / x / mean / rstd / partial");
  const int cvec = C / ch, nthr = (NT / cvec) * cvec;
  const size_t smem = (size_t)nthr * 2 * ch * 4;
  dim3 grid(htrvt_bn_bwd_blocks(npix));
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, grid, dim3(NT), smem, (hipStream_t)stream, (const T*)dy,
                                       (const T*)yact, (const T*)x, mean, rstd, partial, (long long)npix, C, nthr));
  return check_launch("bn_bwd_reduce");
}

extern "C" int htrvt_bn_bwd_finalize(const float* partial, int rows, int C, float count, const float* gamma,
                                     const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef,
                                     void* stream) {
  HTRVT_REQUIRE(partial && gamma && mean && rstd && dgamma && dbeta && coef, "htrvt_bn_bwd_finalize: null argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, rows, C, count,
                     gamma, mean, rstd, dgamma, dbeta, coef);
  return check_launch("bn_bwd_finalize");
}

extern "C" int htrvt_bn_bwd_apply(const void* dy, const void* yact, const void* x, const float* coef, void* dx, void* gout,
                                  int64_t npix, int C, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0, "htrvt_bn_bwd_apply: C=%d unsupported", C);
  const long long nvec = npix * (C / ch);
  dim3 grid(grid_for(nvec));
  if (nvec * 16 >= bn_stream_bytes()) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true>), grid, dim3(NT), (size_t)3 * C * sizeof(float), (hipStream_t)stream, (const T*)dy,
                                         (const T*)yact, (const T*)x, coef, (T*)dx, (T*)gout, nvec, C));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false>), grid, dim3(NT), (size_t)3 * C * sizeof(float), (hipStream_t)stream, (const T*)dy,
                                         (const T*)yact, (const T*)x, coef, (T*)dx, (T*)gout, nvec, C));
  }
  return check_launch("bn_bwd_apply");
}

extern "C" int htrvt_bn_bwd_apply2(const void* g, const void* x1, const float* coef1, void* dx1, const void* x2,
                                   const float* coef2, void* dx2, int64_t npix, int C, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0 && C <= 2048, "htrvt_bn_bwd_apply2: C=%d unsupported", C);
  HTRVT_REQUIRE(g && x1 && x2 && coef1 && coef2 && dx1 && dx2, "htrvt_bn_bwd_apply2: null buffer");
  const long long nvec = npix * (C / ch);
  dim3 grid(grid_for(nvec));
  if (nvec * 16 >= bn_stream_bytes()) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, true>), grid, dim3(NT), (size_t)6 * C * sizeof(float), (hipStream_t)stream, (const T*)g,
                                         (const T*)x1, coef1, (T*)dx1, (const T*)x2, coef2, (T*)dx2, nvec, C));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, false>), grid, dim3(NT), (size_t)6 * C * sizeof(float), (hipStream_t)stream, (const T*)g,
                                         (const T*)x1, coef1, (T*)dx1, (const T*)x2, coef2, (T*)dx2, nvec, C));
  }
  return check_launch("bn_bwd_apply2");
}

extern "C" int htrvt_maxpool_bwd(const void* dpool, const uint8_t* idx, const void* x, const float* scale,
                                 const float* shift, void* g, int B, int H, int W, int C, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0, "htrvt_maxpool_bwd: C=%d unsupported", C);
  HTRVT_REQUIRE((long long)B * W * (C / ch) < (1ll << 31), "htrvt_maxpool_bwd: too many columns");
  dim3 grid((unsigned)(((long long)B * W * (C / ch) + NT - 1) / NT));
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, grid, dim3(NT), 0, (hipStream_t)stream, (const T*)dpool, idx,
                                       (const T*)x, scale, shift, (T*)g, B, H, W, C));
  return check_launch("maxpool_bwd");
}

extern "C" int htrvt_pool_tokens_bwd(const void* dtok, const void* x, const float* keep, void* dx, int B, int H, int N,
                                     int D, int dtype, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  const int Ho = (H - 1) / 2 + 1;
  HTRVT_REQUIRE(D % ch == 0 && N % Ho == 0, "htrvt_pool_tokens_bwd: bad shape");
  const int W = N / Ho;
  if (H == 2) {
    const long long threads = (long long)B * W * (D / ch);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pool_tokens_bwd_h2_kernel<T>, dim3((unsigned)((threads + NT - 1) / NT)), dim3(NT), 0,
                                         (hipStream_t)stream, (const T*)dtok, (const T*)x, keep, (T*)dx, B, W, D));
    return check_launch("pool_tokens_bwd");
  }
  const long long total = (long long)B * H * W * (D / ch);
  dim3 grid(grid_for(total));
  DISPATCH_T(dtype, hipLaunchKernelGGL(pool_tokens_bwd_kernel<T>, grid, dim3(NT), 0, (hipStream_t)stream, (const T*)dtok,
                                       (const T*)x, keep, (T*)dx, B, H, W, D));
  return check_launch("pool_tokens_bwd");
}

extern "C" int htrvt_conv1_wgrad_blocks(int B, int H) {
  int g = B * (H / 2);
  return g > 512 ? 512 : g;
}

extern "C" int htrvt_conv1_wgrad(const void* img, const float* stats, const void* dy, float* dw, float* partial, int B,
                                 int H, int W, int C, int dtype, int img_u8, void* stream) {
  const int ch = dtype == HTRVT_BF16 ? 8 : 4;
  HTRVT_REQUIRE(C % ch == 0 && C / ch <= NT && H % 2 == 0, "htrvt_conv1_wgrad: C=%d unsupported", C);
  const int cvec = C / ch, nthr = (NT / cvec) * cvec;
  const int nblk = htrvt_conv1_wgrad_blocks(B, H);
  size_t smem = (size_t)3 * (W + 2) * 4;
  if (smem < (size_t)nthr * ch * 4) smem = (size_t)nthr * ch * 4;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(conv1_wgrad_kernel<T>, dim3(nblk), dim3(NT), smem, st, img, stats, (const T*)dy,
                                       partial, B, H, W, C, nthr, img_u8));
  hipLaunchKernelGGL(rowsum_f32_kernel, dim3((C * 9 + RS_NT - 1) / RS_NT), dim3(RS_NT), 0, st, partial, nblk, C * 9, dw);
  return check_launch("conv1_wgrad");
}

extern "C" int htrvt_cast_f32(const float* src, void* dst, int64_t n, int dtype, void* stream) {
  HTRVT_REQUIRE(dtype == HTRVT_BF16, "htrvt_cast_f32: only float32 -> bfloat16");
  hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                     (long long)n);
  return check_launch("cast_f32");
}

extern "C" int htrvt_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                           double eps, double weight_decay, int step, void* stream) {
  HTRVT_REQUIRE(n % 4 == 0 && step >= 1, "htrvt_adamw: n must be a multiple of 4 and step >= 1");
  HTRVT_REQUIRE(p && g && m && v, "htrvt_adamw: null buffer");
  float h[HTRVT_ADAMW_SCALARS];
  htrvt_adamw_scalars(lr, beta1, beta2, eps, weight_decay, step, h);
  hipLaunchKernelGGL(adamw_kernel<false>, dim3(grid_for(n / 4)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)(n / 4), h[0], h[1], h[2], h[3], h[4], h[5], h[6], (const float*)nullptr);
  return check_launch("adamw");
}

extern "C" int htrvt_adamw_scalars(double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                                   float* out) {
  HTRVT_REQUIRE(out && step >= 1, "htrvt_adamw_scalars: null output or step < 1");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
  out[0] = (float)(1.0 - lr * weight_decay);
  out[1] = (float)(1.0 - beta1);
  out[2] = (float)beta2;
  out[3] = (float)(1.0 - beta2);
  out[4] = (float)eps;
  out[5] = (float)bc2s;
  out[6] = (float)(-(lr / bc1));
  out[7] = 0.f;
  return 0;
}

extern "C" int htrvt_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev, void* stream) {
  HTRVT_REQUIRE(n % 4 == 0, "htrvt_adamw_dev: n must be a multiple of 4");
  HTRVT_REQUIRE(p && g && m && v && hyper_dev, "htrvt_adamw_dev: null buffer");
  hipLaunchKernelGGL(adamw_kernel<true>, dim3(grid_for(n / 4)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v,
                     (long long)(n / 4), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, hyper_dev);
  return check_launch("adamw_dev");
}
