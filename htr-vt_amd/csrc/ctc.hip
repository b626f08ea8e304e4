// ctc.hip -- fused log-softmax + CTC loss + gradient w.r.t. the logits.
//
// Replaces, in one launch, what model_v1/train.py:21-30 issues as
//   preds.permute(1,0,2).log_softmax(2) ; torch.nn.CTCLoss(reduction='none',
//   zero_infinity=True)(...).mean() ; and their autograd backward
// (ATen's native log-alpha / log-beta / collect kernels, cuDNN disabled).
//
// One workgroup per sample; extended labels l' = [0,l1,0,...,0] (integer, exact)
// sit in LDS, the S = 2L+1 states are spread over the lanes, the T time steps are
// a serial loop with one barrier each.  alpha is spilled to a global workspace
// (read back, L2-hot, by the beta sweep that also forms the gradient
//   d(mean_b nll)/dlogit[b,t,c] = (softmax[b,t,c] - occupancy[b,t,c]) / B ).
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
// the same on the hardware exp2 / log2 units (1 ulp each; the arguments that matter are within a few units of 0):
// this sits on the serial chain of 2 T steps
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  if (m == -INFINITY) return -INFINITY;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

__global__ __launch_bounds__(NT) void ctc_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                 const int* __restrict__ tgt_len, const int* __restrict__ tgt_off,
                                                 float* __restrict__ nll, float* __restrict__ grad,
                                                 float* __restrict__ ws, int T, int C, int Smax, float invB) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* lse = reinterpret_cast<float*>(smem_raw);  // [T]
  int* ext = reinterpret_cast<int*>(lse + T);       // [Smax]
  float* buf0 = reinterpret_cast<float*>(ext + Smax);
  float* buf1 = buf0 + Smax;
  // occupancy of every state at the current step, double buffered over t (one barrier per step), summed per class in a
  // FIXED order after the barrier: the blank through per-wave sums, a label class by walking the list of its states
  // (head / nxt, ascending s).  No LDS float atomics: the gradient is bitwise reproducible.
  float* os0 = buf1 + Smax;   // [Smax]
  float* os1 = os0 + Smax;    // [Smax]
  int* nxt = reinterpret_cast<int*>(os1 + Smax);   // [Smax] next state with the same label, -1 at the end
  int* head = nxt + Smax;     // [C] first state of every label class, -1 if the class does not occur
  float* obw = reinterpret_cast<float*>(head + C);  // [2][NT/64] per-wave blank occupancy
  __shared__ float s_ll;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = tgt_len[b], S = 2 * L + 1;
  const int* lab = targets + tgt_off[b];
  const float* x = logits + (long long)b * T * C;
  float* A = ws + (long long)b * T * Smax;

  for (int s = tid; s < S; s += NT) ext[s] = (s & 1) ? lab[s >> 1] : 0;
  for (int c = tid; c < C; c += NT) head[c] = -1;
  for (int t = wave; t < T; t += NT / 64) {  // log-sum-exp of every frame, one wave per frame
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[(long long)t * C + c]);
    m = wave_max(m);
    float e = 0.f;
    for (int c = lane; c < C; c += 64) e += expf(x[(long long)t * C + c] - m);
    e = wave_sum(e);
    if (lane == 0) lse[t] = m + logf(e);
  }
  __syncthreads();

  // ---- alpha sweep ----
  float* prev = buf0;
  float* cur = buf1;
  for (int s = tid; s < S; s += NT) {
    float a = -INFINITY;
    if (s < 2) a = x[ext[s]] - lse[0];
    prev[s] = a;
    A[s] = a;
  }
  __syncthreads();
  if (S <= NT) {
    // one state per thread: its label, its skip rule and -- one step ahead, so that no step waits for memory -- its
    // emission x[t][label] live in registers; a step is three LDS reads, one log-sum-exp, one barrier
    const bool act = tid < S;
    const int e = act ? ext[tid] : 0;
    const bool skip = act && tid >= 2 && e != 0 && e != ext[tid - 2];
    float xn = (act && T > 1) ? x[(long long)C + e] : 0.f;
    for (int t = 1; t < T; ++t) {
      const float xc = xn;
      if (act && t + 1 < T) xn = x[(long long)(t + 1) * C + e];
      if (act) {
        const float a0 = prev[tid];
        const float a1 = tid >= 1 ? prev[tid - 1] : -INFINITY;
        const float a2 = skip ? prev[tid - 2] : -INFINITY;
        const float a = lse3_fast(a0, a1, a2) + (xc - lse[t]);
        cur[tid] = a;
        A[(long long)t * Smax + tid] = a;
      }
      __syncthreads();
      float* tmp = prev;
      prev = cur;
      cur = tmp;
    }
  } else {
  for (int t = 1; t < T; ++t) {
    const float* xt = x + (long long)t * C;
    const float l = lse[t];
    for (int s = tid; s < S; s += NT) {
      const int e = ext[s];
      const float a0 = prev[s];
      const float a1 = s >= 1 ? prev[s - 1] : -INFINITY;
      const float a2 = (s >= 2 && e != 0 && e != ext[s - 2]) ? prev[s - 2] : -INFINITY;
      const float a = lse3(a0, a1, a2) + (xt[e] - l);
      cur[s] = a;
      A[(long long)t * Smax + s] = a;
    }
    __syncthreads();
    float* tmp = prev;
    prev = cur;
    cur = tmp;
  }
  }
  if (tid == 0) {
    s_ll = lse2(prev[S - 1], S > 1 ? prev[S - 2] : -INFINITY);
    for (int s = S - 2; s >= 1; s -= 2) {   // label states, last to first: every class list ends up ascending in s
      const int e = ext[s];
      if (e > 0 && e < C) {
        nxt[s] = head[e];
        head[e] = s;
      }
    }
  }
  __syncthreads();
  const float ll = s_ll;
  const bool feasible = ll != -INFINITY;
  if (tid == 0) nll[b] = feasible ? -ll : 0.f;
  if (grad == nullptr) return;
  float* g = grad + (long long)b * T * C;
  if (!feasible) {  // zero_infinity: zero loss and zero gradient
    for (int i = tid; i < T * C; i += NT) g[i] = 0.f;
    return;
  }

  // ---- beta sweep + gradient ----
  // prev/cur are reused for beta; all reads of the alpha buffers are behind the barrier above
  if (S <= NT) {
    const bool act = tid < S;
    const int e = act ? ext[tid] : 0;
    const bool skip = act && tid + 2 < S && ext[tid + 2] != 0 && ext[tid + 2] != e;
    float xn = act ? x[(long long)(T - 1) * C + e] : 0.f;          // emission and alpha of the step about to run
    float an = act ? A[(long long)(T - 1) * Smax + tid] : 0.f;
    float gn = tid < C ? x[(long long)(T - 1) * C + tid] : 0.f;    // logit of class `tid` for the gradient row
    // the first four states of class `tid` in registers (their occupancies are then four independent LDS reads per step
    // instead of a dependent walk through the list); `rest` continues the list for a label that occurs more often
    int cs0 = -1, cs1 = -1, cs2 = -1, cs3 = -1, rest = -1;
    if (tid > 0 && tid < C) {
      cs0 = head[tid];
      cs1 = cs0 >= 0 ? nxt[cs0] : -1;
      cs2 = cs1 >= 0 ? nxt[cs1] : -1;
      cs3 = cs2 >= 0 ? nxt[cs2] : -1;
      rest = cs3 >= 0 ? nxt[cs3] : -1;
    }
    for (int t = T - 1; t >= 0; --t) {
      const float* xt = x + (long long)t * C;
      const float l = lse[t];
      float* os = (t & 1) ? os1 : os0;
      float* ow = obw + (t & 1) * (NT / 64);
      const float xc = xn, ac = an, gc = gn;
      if (t > 0) {
        if (act) {
          xn = x[(long long)(t - 1) * C + e];
          an = A[(long long)(t - 1) * Smax + tid];
        }
        if (tid < C) gn = x[(long long)(t - 1) * C + tid];
      }
      if (act) {
        float bt;
        if (t == T - 1) {
          bt = (tid >= S - 2) ? (xc - l) : -INFINITY;
        } else {
          const float b0 = prev[tid];
          const float b1 = tid + 1 < S ? prev[tid + 1] : -INFINITY;
          const float b2 = skip ? prev[tid + 2] : -INFINITY;
          bt = lse3_fast(b0, b1, b2) + (xc - l);
        }
        cur[tid] = bt;
      }
      // occupancy exp(alpha + beta - emission - ll): the blank states (every even s, half of all) are summed per wave
      float ob = 0.f;
      if (act) {
        const float ab = ac + cur[tid];
        const float o = ab != -INFINITY ? __expf(ab - (xc - l) - ll) : 0.f;
        if (e == 0) ob = o;
        else os[tid] = o;
      }
      ob = wave_sum(ob);
      if (lane == 0) ow[wave] = ob;
      __syncthreads();
      for (int c = tid; c < C; c += NT) {
        const float xv = c == tid ? gc : xt[c];
        float oc = 0.f;
        if (c == 0) {
#pragma unroll
          for (int w = 0; w < NT / 64; ++w) oc += ow[w];
        } else if (c == tid) {     // ascending state order, as the list: the sum is the same on every run
          const float o0 = cs0 >= 0 ? os[cs0] : 0.f, o1 = cs1 >= 0 ? os[cs1] : 0.f;
          const float o2 = cs2 >= 0 ? os[cs2] : 0.f, o3 = cs3 >= 0 ? os[cs3] : 0.f;
          oc = ((o0 + o1) + o2) + o3;
          for (int s2 = rest; s2 >= 0; s2 = nxt[s2]) oc += os[s2];
        } else {
          for (int s2 = head[c]; s2 >= 0; s2 = nxt[s2]) oc += os[s2];
        }
        g[(long long)t * C + c] = (expf(xv - l) - oc) * invB;
      }
      float* tmp = prev;
      prev = cur;
      cur = tmp;
    }
    return;
  }
  for (int t = T - 1; t >= 0; --t) {
    const float* xt = x + (long long)t * C;
    const float l = lse[t];
    float* os = (t & 1) ? os1 : os0;
    for (int s = tid; s < S; s += NT) {
      const int e = ext[s];
      float bt;
      if (t == T - 1) {
        bt = (s >= S - 2) ? (xt[e] - l) : -INFINITY;
      } else {
        const float b0 = prev[s];
        const float b1 = s + 1 < S ? prev[s + 1] : -INFINITY;
        const float b2 = (s + 2 < S && ext[s + 2] != 0 && ext[s + 2] != e) ? prev[s + 2] : -INFINITY;
        bt = lse3(b0, b1, b2) + (xt[e] - l);
      }
      cur[s] = bt;
      const float ab = A[(long long)t * Smax + s] + bt;
      os[s] = ab != -INFINITY ? expf(ab - (xt[e] - l) - ll) : 0.f;
    }
    __syncthreads();
    if (wave == NT / 64 - 1) {   // blank class: the even states, lane-strided partial sums + a shuffle tree (fixed order)
      float ob = 0.f;
      for (int s = 2 * lane; s < S; s += 128) ob += os[s];
      ob = wave_sum(ob);
      if (lane == 0) g[(long long)t * C] = (expf(xt[0] - l) - ob) * invB;
    }
    for (int c = tid + 1; c < C; c += NT) {
      float oc = 0.f;
      for (int s2 = head[c]; s2 >= 0; s2 = nxt[s2]) oc += os[s2];
      g[(long long)t * C + c] = (expf(xt[c] - l) - oc) * invB;
    }
    float* tmp = prev;
    prev = cur;
    cur = tmp;
  }
}

}  // namespace

extern "C" size_t htrvt_ctc_workspace_floats(int B, int T, int max_target_len) {
  return (size_t)B * T * (2 * max_target_len + 1);
}

extern "C" int htrvt_ctc_loss(const float* logits, const int32_t* targets, const int32_t* tgt_len, const int32_t* tgt_off,
                              float* nll, float* grad, float* workspace, int B, int T, int C, int max_target_len,
                              float grad_scale, void* stream) {
  HTRVT_REQUIRE(B > 0 && T > 0 && C > 0 && max_target_len >= 0, "htrvt_ctc_loss: bad shape");
  const int Smax = 2 * max_target_len + 1;
  const size_t smem = (size_t)(T + 6 * Smax + C + 2 * (NT / 64)) * 4;
  HTRVT_REQUIRE(smem <= 60 * 1024, "htrvt_ctc_loss: T=%d / target length %d too large for LDS", T, max_target_len);
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(NT), smem, (hipStream_t)stream, logits, targets, tgt_len, tgt_off, nll, grad,
                     workspace, T, C, Smax, grad_scale / (float)B);
  return check_launch("ctc_loss");
}
