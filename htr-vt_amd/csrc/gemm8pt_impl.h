// gemm8pt_impl.h -- the 8-phase bfloat16 GEMM schedule (gemm8p_impl.h) for MN-MAJOR x MN-MAJOR operands (round 4):
//
//   C[m][n] = alpha * sum_k A[k][m] * B[k][n]      float32 C, optionally one K range of a split-K launch into its own slab
//
// i.e. the weight gradients of the Linear layers, dW[out][in] = dy^T x with K = B * N tokens (HTR_VT.py:22-37,76
// backward; 16 launches per step).  They ran on the one-barrier-per-k-tile LDS-DMA kernel in which every wave loads AND
// multiplies (845-885 TFLOP/s, proj 600): here the two wave groups of the 8-phase schedule take turns instead -- one group's
// 16 MFMAs cover the other's fragment reads and DMA issue -- and three half tiles stay in flight behind a counted vmcnt.
//
// LDS image of an MN-major half tile: [64 k][128 mn] bf16 = 64 rows of 256 B; a 1-KiB DMA piece is four k-rows (lane L:
// row L >> 4, 16-byte chunk L & 15).  Fragments are read with ds_read_b64_tr_b16 (hardware transpose: a 16-lane group
// reads 4 k-rows x 16 mn, lane 4q + p supplies row q / columns 4p .. 4p+3 and receives column (lane & 15)), two reads per
// 16 x 32 MFMA operand.  Chunk c of k-row k is stored at c ^ S(k), S(k) = 2 * ((k & 3) | ((k >> 3) & 1) << 2): the eight
// k-rows a 32-lane half touches per read (k = 8 fg + q, fg in {0, 1} or {2, 3}) land on eight different 32-byte bank
// groups -- conflict-free (tools/lds_bank_check.py model) -- and S is the same for both pieces a lane issues.
// No column permutation is needed: with "n on the MFMA rows" a lane's four accumulator registers are four consecutive
// float32 columns = one 16-byte store.
#pragma once
#include "gemm8p_impl.h"

namespace g8 {

__device__ __forceinline__ int tswz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// one MN-major operand: [K][dim] with leading dimension ld (elements); this workgroup's columns start at mn0
struct TLoad {
  unsigned off[2][2];     // [half][piece]: byte offset of the lane's 16 bytes in k-tile 0 (or OOB: column past the matrix)
  int klocal[2];          // the k-row (inside a k-tile) of the lane's two pieces
  unsigned kstep;         // bytes per k-tile
  int k0;                 // first k of the k-tile the next half tile belongs to
  i32x4_t rsrc;

  __device__ __forceinline__ void init(const char* base, long long ld, int mn0, int dim, int kbeg, int wave, int lane) {
    rsrc = make_rsrc(base);
    k0 = kbeg;
    kstep = (unsigned)(BK * ld * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pi = wave + 8 * i;
      const int kl = 4 * pi + (lane >> 4);
      klocal[i] = kl;
      const int cs = (lane & 15) ^ tswz(kl);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mn = mn0 + h * 128 + 8 * cs;
        off[h][i] = mn < dim ? (unsigned)(((long long)(kbeg + kl) * ld + mn) * 2) : OOB;
      }
    }
  }
  template <int H>
  __device__ __forceinline__ void issue(unsigned lds_half, int kend, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned voff = (k0 + klocal[i] < kend) ? off[H][i] : OOB;
      dma16(rsrc, __builtin_amdgcn_readfirstlane(lds_half + (wave + 8 * i) * 1024), voff);
    }
    if constexpr (H == 1) {
      k0 += BK;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) off[h][i] = off[h][i] < OOB ? off[h][i] + kstep : OOB;
    }
  }
};

template <class P>
__device__ __forceinline__ void gemm8pt_body(const P& p, const int block_x) {
  using C = Cfg<256, 2, 4>;
  constexpr int MT = C::MT, NT = C::NT;      // 4 x 2 MFMA tiles per quadrant
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int ntiles = p.tiles_m * p.tiles_n;
  int id;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = block_x & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (block_x >> 3);
  }
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;
  const int wr = wave / C::WARPS_N, wc = wave - wr * C::WARPS_N;

  TLoad la, lb;
  la.init(p.A, p.lda, m0, p.M, kbeg, wave, lane);
  lb.init(p.B, p.ldb, n0, p.N, kbeg, wave, lane);

  f32x4_t acc[2][2][MT][NT];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[x][y][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nkt = (kend - kbeg + BK - 1) / BK;
  const unsigned lds0 = lds_addr_of(smem);
  constexpr unsigned OA0 = 0, OA1 = C::A_HALF, OB0 = 2 * C::A_HALF, OB1 = 2 * C::A_HALF + C::B_HALF;

  // transposed fragment reads: lane (fg = l >> 4, q = (l & 15) >> 2, pp = l & 3); read r of k-step s covers k-rows
  // 32 s + 8 fg + 4 r + q, the lane supplies columns 16 t + 4 pp .. + 3 of its wave's block (8 bytes)
  const int fg = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int ksw = tswz(8 * fg + q);
  unsigned rdA[MT], rdB[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int chunk = (wr * C::SM + 16 * i) / 8 + (pp >> 1);
    rdA[i] = (unsigned)((8 * fg + q) * 256 + ((chunk ^ ksw) << 4) + (pp & 1) * 8);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int chunk = (wc * C::SN + 16 * j) / 8 + (pp >> 1);
    rdB[j] = (unsigned)((8 * fg + q) * 256 + ((chunk ^ ksw) << 4) + (pp & 1) * 8);
  }
  typedef __attribute__((address_space(3))) s16x4_t* lptr;
  auto ldtr = [&](const char* half, unsigned off) -> bf16x8_t {      // k-rows off .. off + 3 rows and + 4 .. + 7 rows
    const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(half + off));
    const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(half + off + 4 * 256));
    const s16x8_t r = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };

  auto stageA = [&](auto xc, auto bufc) {
    constexpr int X = decltype(xc)::value, BUFI = decltype(bufc)::value;
    la.template issue<X>(lds0 + BUFI * C::BUF + (X ? OA1 : OA0), kend, wave);
  };
  auto stageB = [&](auto yc, auto bufc) {
    constexpr int Y = decltype(yc)::value, BUFI = decltype(bufc)::value;
    lb.template issue<Y>(lds0 + BUFI * C::BUF + (Y ? OB1 : OB0), kend, wave);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue (gemm8p_body's): k-tile 0 complete + three half tiles of k-tile 1 ----
  stageB(I0{}, I0{});
  stageA(I0{}, I0{});
  stageB(I1{}, I0{});
  stageA(I1{}, I0{});
  stageB(I0{}, I1{});
  stageA(I0{}, I1{});
  stageB(I1{}, I1{});
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[MT][2], fb0[NT][2], fb1[NT][2];
  auto mma = [&](auto xc, auto yc, bf16x8_t (&fbx)[NT][2]) {
    constexpr int X = decltype(xc)::value, Y = decltype(yc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[X][Y][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbx[j][s], fa[i][s], acc[X][Y][i][j], 0, 0, 0);
  };
  auto readA = [&](const char* half) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      fa[i][0] = ldtr(half, rdA[i]);
      fa[i][1] = ldtr(half, rdA[i] + 32 * 256);
    }
  };
  auto readB = [&](const char* half, bf16x8_t (&f)[NT][2]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f[j][0] = ldtr(half, rdB[j]);
      f[j][1] = ldtr(half, rdB[j] + 32 * 256);
    }
  };
#define G8T_MFMA_PHASE(X, Y, FB)                 \
  __builtin_amdgcn_s_barrier();                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_setprio(1);                 \
  mma(X, Y, FB);                                 \
  __builtin_amdgcn_s_setprio(0);                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();

  auto ktile = [&](auto bufc) {
    constexpr int BUFI = decltype(bufc)::value;
    using BX = std::integral_constant<int, BUFI>;
    using BY = std::integral_constant<int, BUFI ^ 1>;
    const char* base = smem + BUFI * C::BUF;
    // phase 1: b0 (2 reads per fragment: 8), then a0 (16); DMA of A half 1 of k-tile kt+1
    readB(base + OB0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    readA(base + OA0);
    __builtin_amdgcn_sched_barrier(0);
    stageA(I1{}, BY{});
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");   // 24 reads were issued and at most 15 can be outstanding: the 8 b0 reads (issued first) have returned, B half 0 may be restaged next phase
    G8T_MFMA_PHASE(I0{}, I0{}, fb0)
    // phase 2: b1; DMA of B half 0 of k-tile kt+2
    readB(base + OB1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stageB(I0{}, BX{});
    G8T_MFMA_PHASE(I0{}, I1{}, fb1)
    // phase 3: a1; DMA of A half 0 of k-tile kt+2
    readA(base + OA1);
    __builtin_amdgcn_sched_barrier(0);
    stageA(I0{}, BX{});
    G8T_MFMA_PHASE(I1{}, I1{}, fb1)
    // phase 4: DMA of B half 1 of k-tile kt+2; k-tile kt+1 has landed
    stageB(I1{}, BX{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G8T_MFMA_PHASE(I1{}, I0{}, fb0)
  };
  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    ktile(I0{});
    ktile(I1{});
  }
  if (kt < nkt) ktile(I0{});
#undef G8T_MFMA_PHASE
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lane (g = l >> 4, jr = l & 15) holds C[m0 + x*128 + wr*SM + 16 i + jr][n0 + y*128 + wc*SN + 16 j + 4 g .. + 3] ----
  const int g = lane >> 4, jr = lane & 15;
  float* Cf = reinterpret_cast<float*>(p.C) + (p.slab_stride > 0 ? (long long)z * p.slab_stride : 0ll);
  const float alpha = p.alpha;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + x * C::HM + wr * C::SM + 16 * i + jr;
      if (m < p.M) {
        float* row = Cf + (long long)m * p.ldc;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = n0 + y * C::HN + wc * C::SN + 16 * j + 4 * g;
            if (n < p.N) *reinterpret_cast<f32x4_t*>(row + n) = acc[x][y][i][j] * alpha;
          }
      }
    }
}

__global__ __launch_bounds__(512) void gemm8pt_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm8pt_body(*kp, (int)blockIdx.x);
}

inline int launch_t(const KParams& p, int zdim, hipStream_t st) {
  using C = Cfg<256, 2, 4>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8pt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm8pt_kernel, dim3(p.tiles_m * p.tiles_n, 1, zdim), dim3(512), C::LDS_BYTES, st, p);
  set_last_kernel("gemm8pt_kernel");
  const int rc = check_launch("gemm8pt_kernel");
  return rc ? rc : 1;
}

}  // namespace g8
