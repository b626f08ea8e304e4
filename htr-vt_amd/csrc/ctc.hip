// ctc.hip -- fused log-softmax + CTC loss + gradient w.r.t. the logits.
//
// Replaces, in one launch, what model_v1/train.py:21-30 issues as
//   preds.permute(1,0,2).log_softmax(2) ; torch.nn.CTCLoss(reduction='none',
//   zero_infinity=True)(...).mean() ; and their autograd backward
// (ATen's native log-alpha / log-beta / collect kernels, cuDNN disabled).
//
// Extended labels l' = [0,l1,0,...,0] (integer, exact), S = 2L+1 states, T serial time steps.
//
// Targets of up to 127 labels (S <= 256) -- three launches:
//   ctc_lse_kernel    log-sum-exp of every frame, one wave per frame
//   ctc_sweep_kernel  ONE WAVE per (sample, direction): the alpha and the beta recursion of a sample do not depend on
//                     each other, so they run at the same time on different CUs (2 B waves: B = 128 fills the 256 CUs).
//                     A lane keeps K = ceil(S/64) consecutive states in registers and gets its neighbours' edge states
//                     through DPP wave shifts: a time step has no LDS traffic and no barrier; emissions are gathered
//                     four steps ahead.  alpha and beta go to the workspace.
//   ctc_grad_kernel   d(mean_b nll)/dlogit[b,t,c] = (softmax[b,t,c] - occupancy[b,t,c]) / B for all (b, t) rows in
//                     parallel, one wave per row; the occupancies of a class are summed in a FIXED order (class lists
//                     in ascending state order, the blank through a shuffle tree): bitwise reproducible, no atomics.
// Longer targets: ctc_kernel, one workgroup per sample, states over the lanes through LDS, one barrier per step, the
// beta sweep forming the gradient as it goes.
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


// ---------------------------------------------------------------------------------------------
// targets of at most 64 * KMAX / 2 labels.  alpha, beta, the frame log-sum-exp and the log-likelihood are kept in
// BASE-2 log units there (v_exp_f32 / v_log_f32 are base-2: no scaling on the serial chain); rows of alpha / beta have
// 64 K columns, so that every lane stores all its K states unconditionally (inactive ones as -inf) and gathers
// unconditionally -- loads or stores inside divergent branches would force a full s_waitcnt vmcnt(0) per time step.
// ---------------------------------------------------------------------------------------------
constexpr int KMAX = 4;
constexpr int PF = 4;   // emission gathers in flight per state (time steps ahead)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// lane i <- lane i-1 (lane 0: -inf) / lane i <- lane i+1 (lane 63: -inf)
__device__ __forceinline__ float wave_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -INFINITY), __builtin_bit_cast(int, v),
                                                               0x130, 0xf, 0xf, false));
}
// base-2 log-sum-exp, branch-free: all -inf -> exp2(-inf - 0) = 0, log2(0) = -inf.  The sum lies in [1, 3] otherwise:
// the raw hardware instructions need no denormal handling here.
__device__ __forceinline__ float l2se3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  const float mm = m == -INFINITY ? 0.f : m;
  return mm + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - mm) + __builtin_amdgcn_exp2f(b - mm) + __builtin_amdgcn_exp2f(c - mm));
}
__device__ __forceinline__ float l2se2(float a, float b) { return l2se3(a, b, -INFINITY); }

// lse2[row] = log2(sum_c exp(x[row][c])), one wave per frame
__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ logits, float* __restrict__ lse2, int rows, int C) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = logits + (long long)row * C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float e = 0.f;
  for (int c = lane; c < C; c += 64) e += expf(x[c] - m);
  e = wave_sum(e);
  if (lane == 0) lse2[row] = (m + logf(e)) * LOG2E;
}

// blockIdx.x = dirs * b + direction (direction 1 = beta, only when the gradient is wanted: dirs = 2)
template <int K, int dir>
__device__ __forceinline__ void ctc_sweep_body(const float* __restrict__ logits, const int* __restrict__ targets,
                                               const int* __restrict__ tgt_len, const int* __restrict__ tgt_off,
                                               const float* __restrict__ lse_all, float* __restrict__ alpha,
                                               float* __restrict__ beta, float* __restrict__ nll, int T, int C, int b) {
  constexpr int SW = 64 * K;
  __shared__ float fin[2];
  const int lane = threadIdx.x;
  const int L = tgt_len[b], S = 2 * L + 1;
  const int* lab = targets + tgt_off[b];
  const float* x = logits + (long long)b * T * C;
  const float* lse = lse_all + (long long)b * T;
  float* W = (dir ? beta : alpha) + (long long)b * T * SW + lane * K;

  int e[K];
  bool act[K], skip[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int s = lane * K + i;
    act[i] = s < S;
    e[i] = (act[i] && (s & 1)) ? lab[s >> 1] : 0;
    // the transition that jumps over a blank: between two DIFFERENT labels only
    if (dir == 0) skip[i] = act[i] && (s & 1) && s >= 3 && e[i] != 0 && lab[(s >> 1) - 1] != e[i];
    else skip[i] = act[i] && (s & 1) && s + 2 < S && lab[(s >> 1) + 1] != 0 && lab[(s >> 1) + 1] != e[i];
  }
  const int tfirst = dir ? T - 1 : 0, step = dir ? -1 : 1;
  float v[K], xq[PF][K];
  {
    const float l0 = lse[tfirst];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int s = lane * K + i;
      const bool start = dir ? (s >= S - 2) : (s < 2);
      const float em = fmaf(x[(long long)tfirst * C + e[i]], LOG2E, -l0);
      v[i] = (act[i] && start) ? em : -INFINITY;
      W[(long long)tfirst * SW + i] = v[i];
    }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int t = min(max(tfirst + step * (1 + u), 0), T - 1);
#pragma unroll
    for (int i = 0; i < K; ++i) xq[u][i] = x[(long long)t * C + e[i]];
  }
  for (int n0 = 1; n0 < T; n0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int n = n0 + u;
      if (n >= T) break;
      const int t = tfirst + step * n, tn = min(max(t + step * PF, 0), T - 1);
      const float l = lse[t];
      float em[K];
#pragma unroll
      for (int i = 0; i < K; ++i) {
        em[i] = fmaf(xq[u][i], LOG2E, -l);
        xq[u][i] = x[(long long)tn * C + e[i]];
      }
      // edge states of the neighbouring lane: one / two states before this lane's first (alpha), after its last (beta)
      float n1, n2;
      if (dir == 0) {
        n1 = wave_shr1(v[K - 1]);
        n2 = K >= 2 ? wave_shr1(v[K >= 2 ? K - 2 : 0]) : wave_shr1(n1);
      } else {
        n1 = wave_shl1(v[0]);
        n2 = K >= 2 ? wave_shl1(v[K >= 2 ? 1 : 0]) : wave_shl1(n1);
      }
      float nv[K];
#pragma unroll
      for (int i = 0; i < K; ++i) {
        float a1, a2;
        if (dir == 0) {
          a1 = i >= 1 ? v[i >= 1 ? i - 1 : 0] : n1;
          a2 = i >= 2 ? v[i >= 2 ? i - 2 : 0] : (i == 1 ? n1 : n2);
        } else {
          a1 = i + 1 < K ? v[i + 1 < K ? i + 1 : 0] : n1;
          a2 = i + 2 < K ? v[i + 2 < K ? i + 2 : 0] : (i + 2 == K ? n1 : n2);
        }
        const float r = l2se3(v[i], a1, skip[i] ? a2 : -INFINITY) + em[i];
        nv[i] = act[i] ? r : -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < K; ++i) {
        v[i] = nv[i];
        W[(long long)t * SW + i] = v[i];
      }
    }
  }
  if (dir == 0) {   // log-likelihood = lse(alpha[T-1][S-1], alpha[T-1][S-2])
    if (lane == 0) fin[0] = fin[1] = -INFINITY;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int s = lane * K + i;
      if (s == S - 1) fin[0] = v[i];
      if (s == S - 2) fin[1] = v[i];
    }
    __syncthreads();
    if (lane == 0) {
      const float ll2 = l2se2(fin[0], fin[1]);
      nll[b] = ll2 != -INFINITY ? -ll2 * LN2 : 0.f;
    }
  }
}

template <int K>
__global__ __launch_bounds__(64) void ctc_sweep_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                       const int* __restrict__ tgt_len, const int* __restrict__ tgt_off,
                                                       const float* __restrict__ lse_all, float* __restrict__ alpha,
                                                       float* __restrict__ beta, float* __restrict__ nll, int T, int C,
                                                       int dirs) {
  const int b = blockIdx.x / dirs;
  if (blockIdx.x - b * dirs == 0) ctc_sweep_body<K, 0>(logits, targets, tgt_len, tgt_off, lse_all, alpha, beta, nll, T, C, b);
  else ctc_sweep_body<K, 1>(logits, targets, tgt_len, tgt_off, lse_all, alpha, beta, nll, T, C, b);
}

// grid (ceil(T / rows_per_wg), B); 4 waves, wave w takes rows t0 + w, t0 + w + 4, ...
template <int K>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                       const int* __restrict__ tgt_len, const int* __restrict__ tgt_off,
                                                       const float* __restrict__ lse_all, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, float* __restrict__ grad, int T,
                                                       int C, int rows_per_wg, float invB) {
  constexpr int SW = 64 * K;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int* ext = reinterpret_cast<int*>(smem_raw);   // [SW]
  int* nxt = ext + SW;                            // [SW] next state with the same label, -1 at the end
  int* head = nxt + SW;                           // [C] first state of every label class, -1 if the class does not occur
  float* occ = reinterpret_cast<float*>(head + C);   // [4][SW] per wave: occupancy of every label state of its row
  const int b = blockIdx.y, t0 = blockIdx.x * rows_per_wg, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = tgt_len[b], S = 2 * L + 1;
  const int* lab = targets + tgt_off[b];
  const float* x = logits + (long long)b * T * C;
  const float* lse = lse_all + (long long)b * T;
  const float* A = alpha + (long long)b * T * SW;
  const float* Bt = beta + (long long)b * T * SW;
  float* g = grad + (long long)b * T * C;
  const int t1 = min(T, t0 + rows_per_wg);

  const float aT1 = A[(long long)(T - 1) * SW + S - 1], aT2 = S > 1 ? A[(long long)(T - 1) * SW + S - 2] : -INFINITY;
  const float ll2 = l2se2(aT1, aT2);   // the value ctc_sweep_kernel turned into nll[b]
  if (ll2 == -INFINITY) {              // zero_infinity: zero loss and zero gradient
    for (int i = t0 * C + tid; i < t1 * C; i += 256) g[i] = 0.f;
    return;
  }
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? lab[s >> 1] : 0;
  for (int c = tid; c < C; c += 256) head[c] = -1;
  __syncthreads();
  if (tid == 0) {
    for (int s = S - 2; s >= 1; s -= 2) {   // label states, last to first: every class list ends up ascending in s
      const int e = ext[s];
      if (e > 0 && e < C) {
        nxt[s] = head[e];
        head[e] = s;
      }
    }
  }
  __syncthreads();
  int e[K];
#pragma unroll
  for (int i = 0; i < K; ++i) e[i] = (lane + 64 * i < S) ? ext[lane + 64 * i] : 0;
  float* ow = occ + wave * SW;
  for (int t = t0 + wave; t < t1; t += 4) {
    const float l = lse[t];
    const float* xt = x + (long long)t * C;
    float ob = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int s = lane + 64 * i;
      const float ab = A[(long long)t * SW + s] + Bt[(long long)t * SW + s];   // -inf beyond S
      // alpha and beta both carry the emission of (t, s): take one out
      const float o = ab != -INFINITY ? __builtin_amdgcn_exp2f(ab - fmaf(xt[e[i]], LOG2E, -l) - ll2) : 0.f;
      if (e[i] == 0) ob += o;
      else ow[s] = o;
    }
    ob = wave_sum(ob);
    // the rows of a wave are private to it: LDS instructions of one wave complete in order, the fences only pin the
    // compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int c = lane; c < C; c += 64) {
      float oc = 0.f;
      if (c == 0) oc = ob;
      else
        for (int s2 = head[c]; s2 >= 0; s2 = nxt[s2]) oc += ow[s2];
      g[(long long)t * C + c] = (__builtin_amdgcn_exp2f(fmaf(xt[c], LOG2E, -l)) - oc) * invB;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int K>
void launch_fast(const float* logits, const int32_t* targets, const int32_t* tgt_len, const int32_t* tgt_off, float* nll,
                 float* grad, float* ws, int B, int T, int C, float invB, hipStream_t st) {
  constexpr int SW = 64 * K;
  float* alpha = ws;
  float* beta = alpha + (size_t)B * T * SW;
  float* lse = beta + (size_t)B * T * SW;
  hipLaunchKernelGGL(ctc_lse_kernel, dim3((B * T + 3) / 4), dim3(256), 0, st, logits, lse, B * T, C);
  const int dirs = grad ? 2 : 1;
  hipLaunchKernelGGL(ctc_sweep_kernel<K>, dim3(B * dirs), dim3(64), 0, st, logits, targets, tgt_len, tgt_off, lse, alpha, beta,
                     nll, T, C, dirs);
  if (grad) {
    const int rows = 32;
    const size_t smem = (size_t)(2 * SW + C + 4 * SW) * 4;
    hipLaunchKernelGGL(ctc_grad_kernel<K>, dim3((T + rows - 1) / rows, B), dim3(256), smem, st, logits, targets, tgt_len,
                       tgt_off, lse, alpha, beta, grad, T, C, rows, invB);
  }
}

}  // namespace

// alpha [B,T,SW] + beta [B,T,SW] + frame log-sum-exp [B,T]; SW = the states rounded up to whole waves of lanes
extern "C" size_t htrvt_ctc_workspace_floats(int B, int T, int max_target_len) {
  const size_t sw = ((size_t)(2 * max_target_len + 1) + 63) / 64 * 64;
  return (size_t)B * T * (2 * sw + 1);
}

extern "C" int htrvt_ctc_loss(const float* logits, const int32_t* targets, const int32_t* tgt_len, const int32_t* tgt_off,
                              float* nll, float* grad, float* workspace, int B, int T, int C, int max_target_len,
                              float grad_scale, void* stream) {
  HTRVT_REQUIRE(B > 0 && T > 0 && C > 0 && max_target_len >= 0, "htrvt_ctc_loss: bad shape");
  const int Smax = 2 * max_target_len + 1;
  const float invB = grad_scale / (float)B;
  if (Smax <= 64 * KMAX) {
    hipStream_t st = (hipStream_t)stream;
    if (Smax <= 64) launch_fast<1>(logits, targets, tgt_len, tgt_off, nll, grad, workspace, B, T, C, invB, st);
    else if (Smax <= 128) launch_fast<2>(logits, targets, tgt_len, tgt_off, nll, grad, workspace, B, T, C, invB, st);
    else if (Smax <= 192) launch_fast<3>(logits, targets, tgt_len, tgt_off, nll, grad, workspace, B, T, C, invB, st);
    else launch_fast<4>(logits, targets, tgt_len, tgt_off, nll, grad, workspace, B, T, C, invB, st);
    return check_launch("ctc_loss");
  }
  const size_t smem = (size_t)(T + 6 * Smax + C + 2 * (NT / 64)) * 4;
  HTRVT_REQUIRE(smem <= 60 * 1024, "htrvt_ctc_loss: T=%d / target length %d too large for LDS", T, max_target_len);
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(NT), smem, (hipStream_t)stream, logits, targets, tgt_len, tgt_off, nll, grad,
                     workspace, T, C, Smax, invB);
  return check_launch("ctc_loss");
}
