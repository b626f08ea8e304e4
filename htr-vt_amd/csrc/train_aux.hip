// train_aux.hip -- the pieces of the reference's training / validation iteration that sit either side of the
// forward+backward path:
//   * SAM's two-step update (utils/sam.py:15-59) on the trainer's flat float32 buffers: squared gradient norm,
//     the "climb" w + rho g / (|g| + 1e-12) with w saved, and the restore (the AdamW launch follows it);
//   * ModelEma.update (utils/utils.py:158-173) as ONE multi-tensor launch over all state_dict entries
//     (float32 parameters / BatchNorm statistics and the int64 num_batches_tracked counters);
//   * greedy CTC decode (valid.py:40-42 + CTCLabelConverter.decode, utils/utils.py:72-86): arg-max over the
//     classes, drop blanks and repeats, left-pack.
// All HBM-bound streaming kernels.
#include "common.h"

// Every kernel here promises torch's elementwise rounding (one rounding per product / sum): no FMA contraction in this
// file.  (The __f*_rn device functions do NOT promise it on this toolchain: without OCML_BASIC_ROUNDED_OPERATIONS
// __fadd_rn(x, y) is plain `x + y`, open to contraction, and __fsqrt_rn is the approximate native square root; plain
// `/` and sqrtf() are the correctly rounded forms under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt.)
#pragma clang fp contract(off)

using namespace htrvt;

namespace {

constexpr int NT = 256;

inline int grid_for(long long work_items, int cap = 256 * 8) {
  long long g = (work_items + NT - 1) / NT;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ------------------------------------------------------------------ sum of squares (deterministic two-stage)
__global__ __launch_bounds__(NT) void sumsq_partial_kernel(const float* __restrict__ x, long long n4, float* __restrict__ partial) {
  __shared__ float red[8];
  float a = 0.f;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float s = block_sum_256(a, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void sumsq_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  __shared__ double red[NT];
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += NT) a += partial[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)red[0];
}

// ------------------------------------------------------------------ SAM first step: old = w ; w += g * rho / (|g| + 1e-12)
__global__ __launch_bounds__(NT) void sam_first_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ old_p,
                                                       long long n4, float rho, const float* __restrict__ norm_sq) {
  // scale = rho / (|g| + 1e-12) as torch evaluates `float / tensor` (Tensor.__rtruediv__): reciprocal, then multiply
  const float scale = (1.0f / (sqrtf(norm_sq[0]) + 1e-12f)) * rho;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    reinterpret_cast<float4*>(old_p)[i] = pp;
    // e_w = 1.0 * grad * scale, rounded, then added (sam.py:24-25); FMA contraction is off in this file
    pp.x = pp.x + gg.x * scale;
    pp.y = pp.y + gg.y * scale;
    pp.z = pp.z + gg.z * scale;
    pp.w = pp.w + gg.w * scale;
    reinterpret_cast<float4*>(p)[i] = pp;
  }
}

__global__ __launch_bounds__(NT) void copy4_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT)
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
}

// ------------------------------------------------------------------ EMA over a table of tensors
__global__ __launch_bounds__(NT) void ema_update_kernel(const HtrvtEmaEntry* __restrict__ table, float d, float omd) {
  const HtrvtEmaEntry e = table[blockIdx.y];
  if (e.is_int64) {
    long long* ema = reinterpret_cast<long long*>(e.ema);
    const long long* mod = reinterpret_cast<const long long*>(e.model);
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < e.numel; i += (long long)gridDim.x * NT) {
      const float a = (float)ema[i] * d, b = omd * (float)mod[i];
      ema[i] = (long long)(a + b);     // float math, truncating copy_ (utils.py:173 on an int64 entry)
    }
    return;
  }
  float* ema = reinterpret_cast<float*>(e.ema);
  const float* mod = reinterpret_cast<const float*>(e.model);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < e.numel; i += (long long)gridDim.x * NT) {
    const float a = ema[i] * d, b = omd * mod[i];   // ema*d + (1-d)*model, each product rounded (contraction is off)
    ema[i] = a + b;
  }
}

// ------------------------------------------------------------------ greedy CTC decode: one block per sample
__global__ __launch_bounds__(1024) void greedy_decode_kernel(const float* __restrict__ logits, int T, int C, long long ld,
                                                             int ncharacter, int* __restrict__ out, int* __restrict__ out_len) {
  extern __shared__ int sh[];   // [T] arg-max, then [T] keep flags scanned
  int* best = sh;
  int* pos = sh + T;
  const float* lb = logits + (long long)blockIdx.x * T * ld;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float* row = lb + (long long)t * ld;
    float m = row[0];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const float v = row[c];
      if (v > m) {          // first maximum, as torch.max(dim)
        m = v;
        am = c;
      }
    }
    best[t] = am;
  }
  __syncthreads();
  // keep[t] = best != 0 && best != best[t-1] && best < len(character)   (utils.py:80)
  if (threadIdx.x == 0) {
    int n = 0;
    for (int t = 0; t < T; ++t) {
      const int b = best[t];
      const bool keep = b != 0 && !(t > 0 && best[t - 1] == b) && b < ncharacter;
      pos[t] = keep ? n : -1;
      n += keep ? 1 : 0;
    }
    out_len[blockIdx.x] = n;
  }
  __syncthreads();
  int* ob = out + (long long)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    if (pos[t] >= 0) ob[pos[t]] = best[t];
  }
}

}  // namespace

extern "C" int htrvt_sumsq_blocks(int64_t n) { return grid_for(n / 4, 1024); }

extern "C" int htrvt_sumsq(const float* x, int64_t n, float* partial, float* out, void* stream) {
  HTRVT_REQUIRE(n > 0 && n % 4 == 0, "htrvt_sumsq: n must be a positive multiple of 4");
  const int nblk = htrvt_sumsq_blocks(n);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(NT), 0, (hipStream_t)stream, x, (long long)(n / 4), partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, partial, nblk, out);
  return check_launch("sumsq");
}

extern "C" int htrvt_sam_first_step(float* p, const float* g, float* old_p, int64_t n, float rho, const float* norm_sq,
                                    void* stream) {
  HTRVT_REQUIRE(n > 0 && n % 4 == 0 && rho >= 0.f, "htrvt_sam_first_step: n must be a multiple of 4, rho >= 0");
  hipLaunchKernelGGL(sam_first_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, (hipStream_t)stream, p, g, old_p,
                     (long long)(n / 4), rho, norm_sq);
  return check_launch("sam_first_step");
}

extern "C" int htrvt_sam_restore(float* p, const float* old_p, int64_t n, void* stream) {
  HTRVT_REQUIRE(n > 0 && n % 4 == 0, "htrvt_sam_restore: n must be a multiple of 4");
  hipLaunchKernelGGL(copy4_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, (hipStream_t)stream, p, old_p, (long long)(n / 4));
  return check_launch("sam_restore");
}

extern "C" int htrvt_ema_update(const HtrvtEmaEntry* table, int count, int64_t max_numel, double decay, void* stream) {
  HTRVT_REQUIRE(count > 0 && max_numel > 0 && decay >= 0.0 && decay <= 1.0, "htrvt_ema_update: bad arguments");
  dim3 grid(grid_for(max_numel, 64), count);
  hipLaunchKernelGGL(ema_update_kernel, grid, dim3(NT), 0, (hipStream_t)stream, table, (float)decay, (float)(1.0 - decay));
  return check_launch("ema_update");
}

extern "C" int htrvt_ctc_greedy_decode(const float* logits, int B, int T, int C, int64_t ld, int ncharacter, int32_t* out,
                                       int32_t* out_len, void* stream) {
  HTRVT_REQUIRE(B > 0 && T > 0 && C > 0 && ld >= C && T <= 16384, "htrvt_ctc_greedy_decode: bad shape B=%d T=%d C=%d", B, T, C);
  const int nth = T >= 1024 ? 1024 : ((T + 63) / 64) * 64;
  hipLaunchKernelGGL(greedy_decode_kernel, dim3(B), dim3(nth), 2 * T * sizeof(int), (hipStream_t)stream, logits, T, C,
                     (long long)ld, ncharacter, out, out_len);
  return check_launch("ctc_greedy_decode");
}
