// relayout.hip -- the per-step weight re-layouts of the hot path, as ONE launch per direction instead of one per tensor:
//   * float32 master conv weight [Co][Ci][taps]  ->  forward pack [Co][taps][Cpad_i] and dgrad pack [Ci][row_taps][Cpad_o]
//     (the B operands of the implicit-GEMM convolutions, reference resnet18.py:23-39 nn.Conv2d weights),
//   * float32 Linear weight [out][in]            ->  bf16 copy and bf16 transpose (HTR_VT.py:22-37, forward / dgrad B operands),
//   * conv weight-gradient GEMM output [taps][Cpad_i][Co] float32  +=>  the parameter's .grad layout [Co][Ci][taps].
// Every job is cut into 32 x 32 (x taps) tiles that go through LDS, so that both sides see runs of >= 64 contiguous bytes.
// A launch serves a TABLE of jobs (HtrvtRelayoutJob, include/htrvt.h): workgroup b finds its job from the tile prefix with
// one 64-lane compare + ballot, then runs the same tile body the single-tensor entry points run.  Round 2 launched 15 + 17 +
// 15 of these per step at 15-35 us each (0.5 + 0.26 + 0.3 ms of kernel time, most of it launch-shaped: integer divisions by
// run-time tap counts and 4.6 K-element tiles); the table form is three launches per step.
#include "common.h"

using namespace htrvt;

namespace {

constexpr int NT = 256;
constexpr int RT = 32;                       // tile edge along Co and along Ci
constexpr int MAXT = 9;                      // taps held in one LDS tile
constexpr int LU = 4;                        // global loads a thread keeps in flight in the gather loops
constexpr int SMEM_FLOATS = RT * (RT * MAXT + 1);
static_assert(SMEM_FLOATS >= 64 * 65, "the Linear cast+transpose tile (64 x 65 floats) shares the buffer");

template <typename T>
__device__ __forceinline__ void store_pair(T* p, float a, float b);
template <>
__device__ __forceinline__ void store_pair<bf16_t>(bf16_t* p, float a, float b) {
  *reinterpret_cast<unsigned*>(p) = pack_bf16x2(a, b);
}
template <>
__device__ __forceinline__ void store_pair<float>(float* p, float a, float b) {
  *reinterpret_cast<float2*>(p) = make_float2(a, b);
}

// ---- conv weight -> forward / dgrad packs.  TAPS = 0: run-time tap count (1 .. MAXT) ---------------------------------
template <typename T, int TAPS>
__device__ void pack_conv_tile(const HtrvtRelayoutJob& J, int tile, float* smem) {
  const int taps = TAPS ? TAPS : J.taps;
  const int Co = J.d0, Ci = J.d1, cpi = J.cpad_in, cpo = J.cpad_out;
  const int by = tile / J.tiles_x, bx = tile - by * J.tiles_x;
  const int co0 = by * RT, ci0 = bx * RT;
  const int run = RT * taps, ldt = run + 1;
  const int nco = min(RT, Co - co0), nci = min(RT, Ci - ci0);
  const float* w = static_cast<const float*>(J.src);
  T* fwd = static_cast<T*>(J.dst0);
  T* dgr = static_cast<T*>(J.dst1);
  const int lim = nci * taps;                                   // valid floats of one co row of the tile
  // loads from clamped, always-valid addresses and a select behind them, LU at a time: under `if (valid)` every load sat in its
  // own branch with its own wait, one memory round trip per element and thread (36 per tile)
  for (int i0 = threadIdx.x; i0 < RT * run; i0 += NT * LU) {
    float v[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = i0 + u * NT;
      const int r = i / run, c = i - r * run;
      const bool ok = i < RT * run && r < nco && c < lim;
      const float ld = w[ok ? ((long long)(co0 + r) * Ci + ci0) * taps + c : 0ll];
      v[u] = ok ? ld : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = i0 + u * NT;
      const int r = i / run, c = i - r * run;
      if (i < RT * run) smem[r * ldt + c] = v[u];
    }
  }
  __syncthreads();
  // pads (ci in Ci .. cpi-1, co in Co .. cpo-1) are never written: the buffers are zero-initialised by their owner; the odd
  // element of a pair that straddles the end writes the 0 the masked load left in the tile, into a pad column
  if (fwd != nullptr) {
    if ((cpi & 1) == 0) {
      for (int i = threadIdx.x; i < RT * taps * (RT / 2); i += NT) {
        const int r = i / (taps * (RT / 2)), rem = i - r * (taps * (RT / 2));
        const int t = rem / (RT / 2), ci = (rem - t * (RT / 2)) * 2;
        if (r < nco && ci < nci)
          store_pair<T>(fwd + ((long long)(co0 + r) * taps + t) * cpi + ci0 + ci, smem[r * ldt + ci * taps + t], smem[r * ldt + (ci + 1) * taps + t]);
      }
    } else {
      for (int i = threadIdx.x; i < RT * run; i += NT) {
        const int r = i / run, rem = i - r * run;
        const int t = rem / RT, ci = rem - t * RT;
        if (r < nco && ci < nci) fwd[((long long)(co0 + r) * taps + t) * cpi + ci0 + ci] = from_f32<T>(smem[r * ldt + ci * taps + t]);
      }
    }
  }
  if (dgr != nullptr) {
    const int rt = J.row_taps, t0 = J.tap0;
    if ((cpo & 1) == 0) {
      for (int i = threadIdx.x; i < RT * taps * (RT / 2); i += NT) {
        const int ci = i / (taps * (RT / 2)), rem = i - ci * (taps * (RT / 2));
        const int t = rem / (RT / 2), r = (rem - t * (RT / 2)) * 2;
        if (ci < nci && r < nco)
          store_pair<T>(dgr + ((long long)(ci0 + ci) * rt + t0 + t) * cpo + co0 + r, smem[r * ldt + ci * taps + t],
                        r + 1 < RT ? smem[(r + 1) * ldt + ci * taps + t] : 0.f);
      }
    } else {
      for (int i = threadIdx.x; i < RT * run; i += NT) {
        const int ci = i / run, rem = i - ci * run;
        const int t = rem / RT, r = rem - t * RT;
        if (ci < nci && r < nco) dgr[((long long)(ci0 + ci) * rt + t0 + t) * cpo + co0 + r] = from_f32<T>(smem[r * ldt + ci * taps + t]);
      }
    }
  }
}

// ---- wgrad GEMM output [taps][cpi][Co] float32  +=>  grad [Co][Ci][taps] ---------------------------------------------
template <int TAPS>
__device__ void unpack_wgrad_tile(const HtrvtRelayoutJob& J, int tile, float* smem) {
  const int taps = TAPS ? TAPS : J.taps;
  const int Co = J.d0, Ci = J.d1, cpi = J.cpad_in;
  const int by = tile / J.tiles_x, bx = tile - by * J.tiles_x;
  const int co0 = by * RT, ci0 = bx * RT;
  const int run = RT * taps, ldt = run + 1;
  const int nco = min(RT, Co - co0), nci = min(RT, Ci - ci0);
  const float* packed = static_cast<const float*>(J.src);
  float* grad = static_cast<float*>(J.dst0);
  for (int i0 = threadIdx.x; i0 < RT * run; i0 += NT * LU) {
    float v[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = i0 + u * NT;
      const int ci = i / run, rem = i - ci * run;
      const int t = rem / RT, r = rem - t * RT;
      const bool ok = i < RT * run && ci < nci && r < nco;
      const float ld = packed[ok ? ((long long)t * cpi + ci0 + ci) * Co + co0 + r : 0ll];
      v[u] = ok ? ld : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = i0 + u * NT;
      const int ci = i / run, rem = i - ci * run;
      const int t = rem / RT, r = rem - t * RT;
      if (i < RT * run) smem[r * ldt + ci * taps + t] = v[u];
    }
  }
  __syncthreads();
  const int lim = nci * taps;
  for (int i0 = threadIdx.x; i0 < RT * run; i0 += NT * LU) {
    float g[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {      // (element 0 is re-read by the masked lanes, never written by them)
      const int i = i0 + u * NT;
      const int r = i / run, c = i - r * run;
      const bool ok = i < RT * run && r < nco && c < lim;
      g[u] = grad[ok ? ((long long)(co0 + r) * Ci + ci0) * taps + c : 0ll];
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int i = i0 + u * NT;
      const int r = i / run, c = i - r * run;
      if (i < RT * run && r < nco && c < lim) grad[((long long)(co0 + r) * Ci + ci0) * taps + c] = g[u] + smem[r * ldt + c];
    }
  }
}

// ---- Linear weight [rows][cols] float32 -> dst [rows][cols] (optional) and dst_t [cols][ld_t], both T; 64 x 64 tiles --
// dst_t columns rows .. ld_t-1 are zero filled (the head's class count rounded up to a multiple of 8)
template <typename T>
__device__ void cast_transpose_tile(const HtrvtRelayoutJob& J, int tile, float* smem) {
  const int rows = J.d0, cols = J.d1, ld_t = J.cpad_in;
  const int by = tile / J.tiles_x, bx = tile - by * J.tiles_x;
  const int r0 = by * 64, c0 = bx * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 rows of 64 threads
  const float* src = static_cast<const float*>(J.src);
  T* dst = static_cast<T*>(J.dst0);
  T* dst_t = static_cast<T*>(J.dst1);
  for (int rb = ty; rb < 64; rb += (NT / 64) * LU) {
    float v[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int r = r0 + rb + u * (NT / 64), c = c0 + tx;
      const bool ok = r < rows && c < cols;
      const float ld = src[ok ? (long long)r * cols + c : 0ll];
      v[u] = ok ? ld : 0.f;
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int rr = rb + u * (NT / 64), r = r0 + rr, c = c0 + tx;
      smem[rr * 65 + tx] = v[u];
      if (dst != nullptr && r < rows && c < cols) dst[(long long)r * cols + c] = from_f32<T>(v[u]);
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int cc = ty; cc < 64; cc += NT / 64) {
    const int c = c0 + cc, r = r0 + tx;
    if (c < cols && r < ld_t) dst_t[(long long)c * ld_t + r] = from_f32<T>(smem[tx * 65 + cc]);   // rows >= `rows` were read as 0
  }
}

template <typename T>
__device__ __forceinline__ void relayout_tile(const HtrvtRelayoutJob& J, int tile, float* smem) {
  if (J.kind == HTRVT_RELAYOUT_PACK_CONV) {
    if (J.taps == 9)
      pack_conv_tile<T, 9>(J, tile, smem);
    else if (J.taps == 1)
      pack_conv_tile<T, 1>(J, tile, smem);
    else
      pack_conv_tile<T, 0>(J, tile, smem);
  } else if (J.kind == HTRVT_RELAYOUT_UNPACK_WGRAD) {
    if (J.taps == 9)
      unpack_wgrad_tile<9>(J, tile, smem);
    else if (J.taps == 1)
      unpack_wgrad_tile<1>(J, tile, smem);
    else
      unpack_wgrad_tile<0>(J, tile, smem);
  } else {
    cast_transpose_tile<T>(J, tile, smem);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void relayout_one_kernel(const HtrvtRelayoutJob J) {
  __shared__ float smem[SMEM_FLOATS];
  relayout_tile<T>(J, (int)blockIdx.x, smem);
}

template <typename T>
__global__ __launch_bounds__(NT) void relayout_jobs_kernel(const HtrvtRelayoutJob* __restrict__ jobs, int njobs) {
  __shared__ float smem[SMEM_FLOATS];
  __shared__ int which;
  if (threadIdx.x < 64) {     // job = the last one whose first tile is <= this workgroup (tile0 ascending, njobs <= 64)
    const int lane = threadIdx.x;
    const bool le = lane < njobs && jobs[lane].tile0 <= (int)blockIdx.x;
    const unsigned long long m = __ballot(le);
    if (lane == 0) which = __popcll(m) - 1;
  }
  __syncthreads();
  const HtrvtRelayoutJob J = jobs[which];
  __syncthreads();
  relayout_tile<T>(J, (int)blockIdx.x - J.tile0, smem);
}

// The same launch with the table passed BY VALUE in the kernel-argument segment (<= HTRVT_RELAYOUT_ARG_JOBS jobs): nothing
// to upload, nothing to keep alive, no cache to invalidate when a gradient buffer moves -- what the engine uses.
struct RelayoutArgs {
  int njobs;
  int pad_[3];
  HtrvtRelayoutJob j[HTRVT_RELAYOUT_ARG_JOBS];
};
static_assert(sizeof(RelayoutArgs) <= 4096, "kernel-argument segment");

template <typename T>
__global__ __launch_bounds__(NT) void relayout_args_kernel(const RelayoutArgs a) {
  typedef const __attribute__((address_space(4))) RelayoutArgs KA;
  (void)a;
  KA* ka = (KA*)__builtin_amdgcn_kernarg_segment_ptr();      // indexed in place: a by-value copy would live in scratch
  __shared__ float smem[SMEM_FLOATS];
  __shared__ int which;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool le = lane < ka->njobs && ka->j[lane < HTRVT_RELAYOUT_ARG_JOBS ? lane : 0].tile0 <= (int)blockIdx.x;
    const unsigned long long m = __ballot(le);
    if (lane == 0) which = __popcll(m) - 1;
  }
  __syncthreads();
  const int w = which;
  HtrvtRelayoutJob J;
  J.src = ka->j[w].src, J.dst0 = ka->j[w].dst0, J.dst1 = ka->j[w].dst1, J.kind = ka->j[w].kind, J.d0 = ka->j[w].d0, J.d1 = ka->j[w].d1;
  J.taps = ka->j[w].taps, J.cpad_in = ka->j[w].cpad_in, J.cpad_out = ka->j[w].cpad_out, J.row_taps = ka->j[w].row_taps;
  J.tap0 = ka->j[w].tap0, J.tile0 = ka->j[w].tile0, J.tiles_x = ka->j[w].tiles_x;
  __syncthreads();
  relayout_tile<T>(J, (int)blockIdx.x - J.tile0, smem);
}

int job_tiles(HtrvtRelayoutJob* j) {
  if (j->src == nullptr || j->d0 <= 0 || j->d1 <= 0) return -1;
  if (j->kind == HTRVT_RELAYOUT_CAST_TRANSPOSE) {
    if (j->dst1 == nullptr || j->cpad_in < j->d0) return -1;
    j->tiles_x = (j->d1 + 63) / 64;
    return j->tiles_x * ((j->cpad_in + 63) / 64);
  }
  if (j->kind != HTRVT_RELAYOUT_PACK_CONV && j->kind != HTRVT_RELAYOUT_UNPACK_WGRAD) return -1;
  if (j->taps < 1 || j->taps > MAXT || j->cpad_in < j->d1) return -1;
  if (j->kind == HTRVT_RELAYOUT_PACK_CONV) {
    if (j->dst1 != nullptr && (j->cpad_out < j->d0 || j->tap0 < 0 || j->tap0 + j->taps > j->row_taps)) return -1;
    if (j->dst0 == nullptr && j->dst1 == nullptr) return -1;
  } else if (j->dst0 == nullptr) {
    return -1;
  }
  j->tiles_x = (j->d1 + RT - 1) / RT;
  return j->tiles_x * ((j->d0 + RT - 1) / RT);
}

int launch_one(HtrvtRelayoutJob& j, int dtype, void* stream, const char* what) {
  const int tiles = job_tiles(&j);
  HTRVT_REQUIRE(tiles > 0, "%s: bad arguments", what);
  j.tile0 = 0;
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(relayout_one_kernel<bf16_t>, dim3(tiles), dim3(NT), 0, (hipStream_t)stream, j);
  else
    hipLaunchKernelGGL(relayout_one_kernel<float>, dim3(tiles), dim3(NT), 0, (hipStream_t)stream, j);
  return check_launch(what);
}

}  // namespace

// host side: fills tile0 / tiles_x of every job, returns the number of workgroups of the launch (< 0: bad job)
extern "C" int htrvt_relayout_plan(HtrvtRelayoutJob* jobs, int njobs) {
  HTRVT_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= HTRVT_RELAYOUT_MAX_JOBS, "htrvt_relayout_plan: 1..%d jobs", HTRVT_RELAYOUT_MAX_JOBS);
  int total = 0;
  for (int i = 0; i < njobs; ++i) {
    const int t = job_tiles(&jobs[i]);
    HTRVT_REQUIRE(t > 0, "htrvt_relayout_plan: job %d is malformed (kind %d, %d x %d, taps %d)", i, jobs[i].kind, jobs[i].d0, jobs[i].d1, jobs[i].taps);
    jobs[i].tile0 = total;
    total += t;
  }
  return total;
}

// jobs_dev: the planned table in device memory; total_tiles: what htrvt_relayout_plan returned for it
extern "C" int htrvt_relayout(const HtrvtRelayoutJob* jobs_dev, int njobs, int total_tiles, int dtype, void* stream) {
  HTRVT_REQUIRE(jobs_dev != nullptr && njobs >= 1 && njobs <= HTRVT_RELAYOUT_MAX_JOBS && total_tiles >= 1, "htrvt_relayout: bad table");
  HTRVT_REQUIRE(dtype == HTRVT_BF16 || dtype == HTRVT_F32, "htrvt_relayout: dtype");
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(relayout_jobs_kernel<bf16_t>, dim3(total_tiles), dim3(NT), 0, (hipStream_t)stream, jobs_dev, njobs);
  else
    hipLaunchKernelGGL(relayout_jobs_kernel<float>, dim3(total_tiles), dim3(NT), 0, (hipStream_t)stream, jobs_dev, njobs);
  return check_launch("relayout");
}

// jobs: a PLANNED table in HOST memory (htrvt_relayout_plan), njobs <= HTRVT_RELAYOUT_ARG_JOBS; it is copied into the launch
extern "C" int htrvt_relayout_host(const HtrvtRelayoutJob* jobs, int njobs, int total_tiles, int dtype, void* stream) {
  HTRVT_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= HTRVT_RELAYOUT_ARG_JOBS && total_tiles >= 1, "htrvt_relayout_host: 1..%d planned jobs",
                HTRVT_RELAYOUT_ARG_JOBS);
  HTRVT_REQUIRE(dtype == HTRVT_BF16 || dtype == HTRVT_F32, "htrvt_relayout_host: dtype");
  RelayoutArgs a = {};
  a.njobs = njobs;
  for (int i = 0; i < njobs; ++i) a.j[i] = jobs[i];
  HTRVT_REQUIRE(a.j[0].tile0 == 0 && a.j[njobs - 1].tile0 < total_tiles, "htrvt_relayout_host: the table is not planned");
  if (dtype == HTRVT_BF16)
    hipLaunchKernelGGL(relayout_args_kernel<bf16_t>, dim3(total_tiles), dim3(NT), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(relayout_args_kernel<float>, dim3(total_tiles), dim3(NT), 0, (hipStream_t)stream, a);
  return check_launch("relayout");
}

extern "C" int htrvt_pack_conv_weight_slots(const float* w, void* fwd, void* dgrad, int Co, int Ci, int taps, int cpad_in,
                                            int cpad_out, int row_taps, int tap0, int dtype, void* stream) {
  HtrvtRelayoutJob j = {};
  j.src = w, j.dst0 = fwd, j.dst1 = dgrad, j.kind = HTRVT_RELAYOUT_PACK_CONV;
  j.d0 = Co, j.d1 = Ci, j.taps = taps, j.cpad_in = cpad_in, j.cpad_out = cpad_out, j.row_taps = row_taps, j.tap0 = tap0;
  return launch_one(j, dtype, stream, "pack_conv_weight");
}

extern "C" int htrvt_pack_conv_weight(const float* w, void* fwd, void* dgrad, int Co, int Ci, int taps, int cpad_in,
                                      int cpad_out, int dtype, void* stream) {
  return htrvt_pack_conv_weight_slots(w, fwd, dgrad, Co, Ci, taps, cpad_in, cpad_out, taps, 0, dtype, stream);
}

extern "C" int htrvt_unpack_conv_wgrad(const float* packed, float* grad, int Co, int Ci, int taps, int cpad_in,
                                       void* stream) {
  HtrvtRelayoutJob j = {};
  j.src = packed, j.dst0 = grad, j.kind = HTRVT_RELAYOUT_UNPACK_WGRAD;
  j.d0 = Co, j.d1 = Ci, j.taps = taps, j.cpad_in = cpad_in;
  return launch_one(j, HTRVT_F32, stream, "unpack_conv_wgrad");
}

extern "C" int htrvt_cast_transpose_f32(const float* src, void* dst, void* dst_t, int rows, int cols, int ld_t, int dtype,
                                        void* stream) {
  HTRVT_REQUIRE(dtype == HTRVT_BF16, "htrvt_cast_transpose_f32: only float32 -> bfloat16");
  HtrvtRelayoutJob j = {};
  j.src = src, j.dst0 = dst, j.dst1 = dst_t, j.kind = HTRVT_RELAYOUT_CAST_TRANSPOSE;
  j.d0 = rows, j.d1 = cols, j.cpad_in = ld_t;
  return launch_one(j, dtype, stream, "cast_transpose_f32");
}
