// variants.hip -- the relative-position bias of the window-attention fork (SURVEY 8(f-4);
// /root/reference/model_window/model/HTR_VT.py:23-31 table + index, :45-46 lookup, :113-154 Block._attend: zero padding to
// a multiple of the window, cyclic shift, 1-D window partition, key_padding_mask) as ONE dense additive score bias for
// the attention kernels of the hot path, and its backward:
//   bias[h][i][j]  = inside(i, j) ? table[slot(j) - slot(i) + P - 1][h] : -1e30      (i, j: tokens of the ORIGINAL order)
//   dtable[e][h]   = sum over the pairs (i, j) with inside(i, j) and entry e of dbias[h][i][j]
// Token t sits at position (t - shift) mod Np of the rolled, padded sequence (Np = N rounded up to the window), window =
// position / ws, slot = position % ws; window <= 0: full attention, entry = (j - i) + P - 1.  For a fixed query i and a
// fixed table entry there is at most ONE key j, so the backward is a gather-sum in a fixed order: no atomics.
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT = 256;
constexpr float MASKED = -1.0e30f;

struct WinGeo {
  int N, Np, P, ws, shift;
  __device__ __forceinline__ int pos(int t) const {
    int p = t - shift;
    p %= Np;
    return p < 0 ? p + Np : p;
  }
};

__global__ __launch_bounds__(NT) void relpos_bias_kernel(const float* __restrict__ table, float* __restrict__ bias, WinGeo g,
                                                         int heads, int ldb) {
  const int i = blockIdx.x, h = blockIdx.y;
  float* row = bias + ((long long)h * ldb + i) * ldb;
  if (i >= g.N) {   // padding query rows (the caller pads the sequence for the kernels): anything finite; masked like the columns
    for (int j = threadIdx.x; j < ldb; j += NT) row[j] = j == 0 ? 0.f : MASKED;
    return;
  }
  int wi = 0, si = 0;
  if (g.ws > 0) {
    const int p = g.pos(i);
    wi = p / g.ws;
    si = p - wi * g.ws;
  }
  for (int j = threadIdx.x; j < ldb; j += NT) {
    float v = MASKED;
    if (j < g.N) {
      if (g.ws <= 0) {
        v = table[(long long)(j - i + g.P - 1) * heads + h];
      } else {
        const int p = g.pos(j);
        const int wj = p / g.ws, sj = p - wj * g.ws;
        if (wj == wi) v = table[(long long)(sj - si + g.P - 1) * heads + h];
      }
    }
    row[j] = v;
  }
}

__global__ __launch_bounds__(NT) void relpos_bias_bwd_kernel(const float* __restrict__ dbias, float* __restrict__ dtable, WinGeo g,
                                                             int heads, int ldb) {
  __shared__ float red[NT / 64];
  const int e = blockIdx.x, h = blockIdx.y;
  const int d = e - (g.P - 1);          // slot(j) - slot(i), or j - i
  float acc = 0.f;
  for (int i = threadIdx.x; i < g.N; i += NT) {
    int j = -1;
    if (g.ws <= 0) {
      j = i + d;
    } else {
      const int p = g.pos(i);
      const int wi = p / g.ws, si = p - wi * g.ws;
      const int sj = si + d;
      if (sj >= 0 && sj < g.ws) {
        int t = wi * g.ws + sj + g.shift;      // position -> token: (position + shift) mod Np
        t %= g.Np;
        j = t < 0 ? t + g.Np : t;
      }
    }
    if (j >= 0 && j < g.N) acc += dbias[((long long)h * ldb + i) * ldb + j];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dtable[(long long)e * heads + h] = red[0] + red[1] + red[2] + red[3];
}

int check_geo(int N, int P, int window, int shift, int heads, int ldb, WinGeo* g) {
  HTRVT_REQUIRE(N > 0 && P >= N && heads > 0 && ldb >= N, "relpos_bias: need 0 < N <= num_patches, ld >= N (N=%d P=%d ld=%d)", N, P, ldb);
  HTRVT_REQUIRE(window <= 0 || (shift >= 0 && shift < window), "relpos_bias: shift %d outside [0, window %d)", shift, window);
  g->N = N;
  g->P = P;
  g->ws = window > 0 ? window : 0;
  g->shift = window > 0 ? shift : 0;
  g->Np = window > 0 ? (N + window - 1) / window * window : N;
  return 0;
}

}  // namespace

extern "C" int htrvt_relpos_bias_fwd(const float* table, float* bias, int N, int num_patches, int window, int shift, int heads,
                                     int ld, void* stream) {
  HTRVT_REQUIRE(table && bias, "htrvt_relpos_bias_fwd: null argument");
  WinGeo g;
  if (check_geo(N, num_patches, window, shift, heads, ld, &g)) return -1;
  hipLaunchKernelGGL(relpos_bias_kernel, dim3(ld, heads), dim3(NT), 0, (hipStream_t)stream, table, bias, g, heads, ld);
  return check_launch("relpos_bias_fwd");
}

extern "C" int htrvt_relpos_bias_bwd(const float* dbias, float* dtable, int N, int num_patches, int window, int shift, int heads,
                                     int ld, void* stream) {
  HTRVT_REQUIRE(dbias && dtable, "htrvt_relpos_bias_bwd: null argument");
  WinGeo g;
  if (check_geo(N, num_patches, window, shift, heads, ld, &g)) return -1;
  hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3(2 * num_patches - 1, heads), dim3(NT), 0, (hipStream_t)stream, dbias, dtable, g, heads, ld);
  return check_launch("relpos_bias_bwd");
}
