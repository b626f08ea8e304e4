// gemm_halo_impl.h -- 3x3, stride-1, pad-1 convolution forward / dgrad as implicit GEMM with a HALO-STAGED A operand
// (resnet18.py:26-31: nine of the twelve 3x3 convolutions of the stem, forward and input gradient).
//
// The generic gather kernel (gemm_dma_impl.h, GATHER 1 / 2) stages, for each of the nine taps and each 64-channel
// chunk, its own 256 x 64 A tile: the pixels of the three taps of one kernel row are the same pixels shifted by one
// column, so two thirds of the A bytes that cross the (per-CU, ~50-70 GB/s) LDS-DMA path are re-reads -- and that path,
// not the matrix pipe, bounds the k-tile (2 450 cycles per k-tile against 1 536 for the MFMAs).
// Here an M tile is 256 consecutive pixels of ONE image row (host check: W % 256 == 0); per (kernel row, channel chunk)
// ONE halo tile of 258 pixels x 64 channels is staged and serves three k-tiles, the MFMA waves reading their A
// fragments `shift` rows further down for the next tap.  Operand bytes per three k-tiles: 40 KB (33 real) + 3 x 24 KB
// instead of 3 x 56 KB.  The freed LDS pays for a third B stage: B runs two k-tiles ahead, the halo tile a whole group
// ahead, and every wait is a counted vmcnt that leaves the newest k-tile's pieces in flight.
//
// Structure otherwise as gemm_dma_kernel<256, BN, .., SPEC = 1>: 8 MFMA waves (4 x 2, v_mfma_f32_32x32x16_bf16) + 4
// loader waves, one barrier per k-tile, the LDS-staged bf16 epilogue (BatchNorm column sums; backward-of-ReLU mask and
// BatchNorm-backward sums on dgrad) shared with that kernel.
#pragma once
#include "gemm_dma_impl.h"

namespace {

template <int BN>
struct HaloGeo {
  static constexpr int BM = 256;
  static constexpr int A_PIECES = 40;                    // 320 LDS rows of 128 B: rows 0..257 are pixels w0-1 .. w0+256, the rest zero fill
  static constexpr int A_STAGE = A_PIECES * 1024;
  static constexpr int B_STAGE = Geo<BN>::BYTES;
  static constexpr int NP_B = B_STAGE / 1024 / 4;        // B pieces per loader wave and k-tile
  static constexpr int NP_AH = A_PIECES / 2 / 4;         // A pieces per loader wave in each of a group's first two k-tiles
  static constexpr unsigned B_BASE = 2 * A_STAGE;
  static constexpr int LDS_BYTES = 2 * A_STAGE + 3 * B_STAGE;
  static_assert(B_STAGE % 4096 == 0, "B pieces divide over the four loader waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// DGRAD = false: A rows = output pixels, source = x [B,H,W,Ci];  true: A rows = input pixels, source = dy [B,H,W,Co]
template <int BN, bool DGRAD, class P>
__device__ __forceinline__ void gemm_halo_body(const P& p, const int block_x) {
  using H = HaloGeo<BN>;
  constexpr int BM = 256, NWC = 8, NW_TOTAL = 12, TM = 2, TN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int ntiles = p.tiles_m * p.tiles_n;
  int id = block_x;
  if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Hh = DGRAD ? p.Ho : p.Hi, Ww = DGRAD ? p.Wo : p.Wi, Cs = DGRAD ? p.Co : p.Ci;   // gathered tensor [B,Hh,Ww,Cs]
  const int rowi = m0 / Ww, w0 = m0 - rowi * Ww;      // row (b * Hm + h) of the M index space and first column of this tile
  // forward with a row stride (sh = 2, W stride 1: the first conv of layer 1): M rows are OUTPUT rows, Hm = Ho of them per
  // image, and kernel row gdy of output row h reads input row h * sh + gdy - 1; dgrad is served at stride 1 only
  const int Hm = DGRAD ? Hh : p.Ho;
  const int bimg = rowi / Hm, hrow = rowi - bimg * Hm;
  const int NC = p.Cpad / BK;                           // 64-channel chunks per tap
  const int NG = 3 * NC;                                // groups = (kernel row, chunk); k-tiles = 3 * NG

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < NWC;
  const int lw = (wave - NWC) & 3;
  const int wm = (wave >> 1) & 3, wn = wave & 1;
  if (!consumer) __builtin_amdgcn_s_setprio(3);

  // ---- loader state ----
  DmaLoader<BN, HTRVT_KMAJOR, 0, 4> lb;
  lb.init(p, p.B, p.ldb, n0, p.N, lw, lane);
  const unsigned long long ba = (unsigned long long)p.A;
  const i32x4_t rsrcA = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
  // this lane's first halo row (piece lw) and source chunk; piece lw + 4 i is 32 i rows further down, same swizzle
  const int rho0 = lw * 8 + (lane >> 3);
  const int cgA = (lane & 7) ^ Geo<BM>::swz(rho0);
  const unsigned lane_off = (unsigned)((w0 - 1 + rho0) * Cs + cgA * 8) * 2u;   // may wrap for w0 - 1 + rho0 < 0: masked below
  const unsigned lds0 = lds_addr_of(smem);

  // A halo half `half` (pieces 20 half .. 20 half + 19) of group (gdy, gcc) into A stage `ast`; gvalid false: zero fill
  auto issueA = [&](int half, int ast, int gdy, int gcc, bool gvalid) {
    const int hh = DGRAD ? hrow + 1 - gdy : hrow * p.sh + gdy - 1;
    const bool rowok = gvalid && (unsigned)hh < (unsigned)Hh;
    const unsigned gbase = (unsigned)(((bimg * Hh + hh) * Ww) * Cs + gcc * BK) * 2u;
    const bool chok = gcc * BK + cgA * 8 < Cs;
#pragma unroll
    for (int i = 0; i < H::NP_AH; ++i) {
      const int ii = half * H::NP_AH + i;
      const int rho = rho0 + 32 * ii;
      const int w = w0 - 1 + rho;
      const bool v = rowok && chok && rho < 258 && (unsigned)w < (unsigned)Ww;
      const unsigned voff = v ? gbase + lane_off + (unsigned)(32 * ii * Cs) * 2u : OOB;
      dma16(rsrcA, __builtin_amdgcn_readfirstlane(lds0 + ast * H::A_STAGE + (lw + 4 * ii) * 1024), voff);
    }
  };
  // B tile of k-tile (gdy, gcc, dx) into B stage `bst`
  auto issueB = [&](int bst, int gdy, int gcc, int dx, bool gvalid) {
    const int k0 = gvalid ? (gdy * 3 + dx) * p.Cpad + gcc * BK : p.K;      // >= K: zero fill
    lb.template issue<true>(p, lds0 + H::B_BASE + bst * H::B_STAGE, k0, p.K, lw);
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: halo tile of group 0, B of k-tiles 0 and 1 ----
  if (!consumer) {
    issueA(0, 0, 0, 0, true);
    issueA(1, 0, 0, 0, true);
    issueB(0, 0, 0, 0, true);
    issueB(1, 0, 0, 1, true);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
  }
  __builtin_amdgcn_s_barrier();

  // consumer fragment addressing: A row = wm*64 + i*32 + (lane & 31) + shift
  const int arow = wm * 64 + (lane & 31), ah = lane >> 5;
  auto compute = [&](const char* sa, const char* sb, int shift) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = arow + i * 32 + shift;
        const int chunk = 2 * s + ah;
        fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa + row * 128 + ((chunk ^ Geo<BM>::swz(row)) << 4)));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, HTRVT_KMAJOR>(sb, wn * TN + j, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  int gdy = 0, gcc = 0;           // group g
  for (int g = 0; g < NG; ++g) {
    int ndy = gdy, ncc = gcc + 1;   // group g + 1
    if (ncc == NC) {
      ncc = 0;
      ++ndy;
    }
    const bool nvalid = g + 1 < NG;
    const char* sa = smem + (g & 1) * H::A_STAGE;
    const int nast = (g + 1) & 1;
    // ---- k-tile 3g (dx = 0): B(3g+2) = (g, dx 2) -> stage 2; first half of halo(g+1) ----
    if (!consumer) {
      issueB(2, gdy, gcc, 2, true);
      issueA(0, nast, ndy, ncc, nvalid);
    } else {
      compute(sa, smem + H::B_BASE + 0 * H::B_STAGE, DGRAD ? 2 : 0);
    }
    if (!consumer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NP_AH) : "memory");
    __builtin_amdgcn_s_barrier();
    // ---- k-tile 3g+1 (dx = 1): B(3g+3) = (g+1, dx 0) -> stage 0; second half of halo(g+1) ----
    if (!consumer) {
      issueB(0, ndy, ncc, 0, nvalid);
      issueA(1, nast, ndy, ncc, nvalid);
    } else {
      compute(sa, smem + H::B_BASE + 1 * H::B_STAGE, 1);
    }
    if (!consumer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NP_AH) : "memory");
    __builtin_amdgcn_s_barrier();
    // ---- k-tile 3g+2 (dx = 2): B(3g+4) = (g+1, dx 1) -> stage 1 ----
    if (!consumer) {
      issueB(1, ndy, ncc, 1, nvalid);
    } else {
      compute(sa, smem + H::B_BASE + 2 * H::B_STAGE, DGRAD ? 0 : 2);
    }
    if (!consumer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
    __builtin_amdgcn_s_barrier();
    gdy = ndy;
    gcc = ncc;
  }
  if (!consumer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill pieces issued past the last k-tile
  __builtin_amdgcn_s_barrier();

  epilogue_staged<TN, BN, BM, NW_TOTAL, DGRAD, !DGRAD>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, consumer);
}

template <int BN, bool DGRAD>
__global__ __launch_bounds__(768) void gemm_halo_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_halo_body<BN, DGRAD>(*kp, (int)blockIdx.x);
}

template <int BN, bool DGRAD>
int launch_halo(const KParams& p, hipStream_t st) {
  using H = HaloGeo<BN>;
  static bool attr_done = false;
  auto kern = gemm_halo_kernel<BN, DGRAD>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, H::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", H::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(768), H::LDS_BYTES, st, p);
  set_last_kernel("gemm_halo_kernel<%d, %s>", BN, DGRAD ? "true" : "false");
  const int rc = check_launch("gemm_halo_kernel");
  return rc ? rc : 1;
}

}  // namespace
