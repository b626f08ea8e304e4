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


// One k-tile (64 deep) of a wave's 64 x (32 TN) block from a halo A stage and a K-major B stage, on
// v_mfma_f32_16x16x32_bf16: 4 x (2 TN) tiles of 16 x 16, two k-steps of 32.  Operand lane map: row l & 15, 16-byte k chunk
// 4 s + (l >> 4).  Measured against the v_mfma_f32_32x32x16_bf16 form of the same loop (same bytes from LDS, same
// cycles per FLOP): +2.4 ... +6 % on every stride-1 3x3 convolution of the stem, forward and dgrad
// (profiles/r05_experiments.md) -- the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS give-back,
// item 7).  -DHTRVT_HALO_MFMA32 builds the former form (A/B runs).
typedef float f32x4h_t __attribute__((ext_vector_type(4)));
template <int BN, int TM, int TN>
__device__ __forceinline__ void halo_mma16(f32x4h_t (&a4)[2 * TM][2 * TN], const char* sa, const char* sb, int shift, int wm, int wn,
                                           int lane) {
  const int arow16 = wm * 64 + (lane & 15), ag = lane >> 4;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8_t fa[2 * TM], fb[2 * TN];
    const int chunk = 4 * s + ag;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) {
      const int row = arow16 + i * 16 + shift;
      fa[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa + row * 128 + ((chunk ^ Geo<256>::swz(row)) << 4)));
    }
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      const int row = wn * TN * 32 + j * 16 + (lane & 15);
      fb[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + row * 128 + ((chunk ^ Geo<BN>::swz(row)) << 4)));
    }
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j)
        a4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], a4[i][j], 0, 0, 0);
  }
}
// the four 16 x 16 tiles of 32 x 32 block (i, j) as registers 4 (2 a + b) + r of acc[i][j] (epilogue_staged<..., MF = 16>)
template <int TM, int TN>
__device__ __forceinline__ void halo_acc16_to_32(f32x16_t (&acc)[TM][TN], const f32x4h_t (&a4)[2 * TM][2 * TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][4 * (2 * a + b) + r] = a4[2 * i + a][2 * j + b][r];
}

// float32 C straight from the 16 x 16 accumulator tiles (the split-bf16 parity path, csrc/split.hip: its convolutions are bf16
// products over three times the channels with float32 results; until round 5 they ran on the generic gather because the halo
// kernels only had the LDS-staged bf16 epilogue).  a4[i][j][r] = C(row 16 i + 4 (lane >> 4) + r, column 16 j + (lane & 15)) of the
// wave's 64 x (32 TN) block: plain stores (16 lanes = 64 contiguous bytes of one row), alpha, an optional float32 residual (dgrad),
// and the per-tile BatchNorm column sums of the forward (wave partials through LDS, summed over the four wave rows in a fixed
// order).  Every wave of the workgroup calls it (loader waves with active = false: they only join the barrier).
template <int BN, int TM, int TN, int NTH, bool CSTATS, class P>
__device__ __forceinline__ void halo_epilogue_f32(const f32x4h_t (&a4)[2 * TM][2 * TN], const P& p, int m0, int n0, int wm, int wn,
                                                  int tile_m, int lane, char* smem, bool active) {
  float* const Cf = reinterpret_cast<float*>(p.C);
  const float* const Rf = reinterpret_cast<const float*>(p.residual);
  const int lr = lane & 15, lg = lane >> 4;
  float s1[2 * TN], s2[2 * TN];
#pragma unroll
  for (int j = 0; j < 2 * TN; ++j) s1[j] = s2[j] = 0.f;
  if (active) {
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
      const int n = n0 + wn * TN * 32 + 16 * j + lr;
      const bool nok = n < p.N;
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + 16 * i + 4 * lg + r;
          const float a = a4[i][j][r];       // rows >= M and columns >= N hold exact zeros (zero-filled operands)
          if constexpr (CSTATS) {
            s1[j] += a;
            s2[j] = fmaf(a, a, s2[j]);
          }
          if (nok && m < p.M) {
            const long long o = (long long)m * p.ldc + n;
            float v = a * p.alpha;
            if (Rf != nullptr) v += Rf[o];
            Cf[o] = v;
          }
        }
      }
    }
  }
  if constexpr (CSTATS) {
    if (p.colstats != nullptr) {      // workgroup-uniform
      float* red = reinterpret_cast<float*>(smem);     // [4 wave rows][BN][2]; the operand stages are dead
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) {
        float a = s1[j], q = s2[j];
        a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
        a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
        if (lane < 16 && active) {
          const int c = wn * TN * 32 + 16 * j + lane;
          red[(wm * BN + c) * 2 + 0] = a;
          red[(wm * BN + c) * 2 + 1] = q;
        }
      }
      __syncthreads();
      for (int c = threadIdx.x; c < BN; c += NTH) {
        const int n = n0 + c;
        if (n < p.N) {
          float a = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            a += red[(w * BN + c) * 2];
            q += red[(w * BN + c) * 2 + 1];
          }
          float* dst = p.colstats + (long long)tile_m * 2 * p.N;
          dst[n] = a;
          dst[p.N + n] = q;
        }
      }
    }
  }
}

// DGRAD = false: A rows = output pixels, source = x [B,H,W,Ci];  true: A rows = input pixels, source = dy [B,H,W,Co]
template <int BN, bool DGRAD, class P>
__device__ __forceinline__ void gemm_halo_body_oneloop(const P& p, const int block_x) {
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
#ifndef HTRVT_HALO_MFMA32
  f32x4h_t a4[2 * TM][2 * TN];
#pragma unroll
  for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) a4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
  auto compute = [&](const char* sa, const char* sb, int shift) { halo_mma16<BN, TM, TN>(a4, sa, sb, shift, wm, wn, lane); };
#else
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
#endif

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

#ifndef HTRVT_HALO_MFMA32
  halo_acc16_to_32<TM, TN>(acc, a4);
  epilogue_staged<TN, BN, BM, NW_TOTAL, DGRAD, !DGRAD, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, consumer);
#else
  epilogue_staged<TN, BN, BM, NW_TOTAL, DGRAD, !DGRAD>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, consumer);
#endif
}

// ---------------------------------------------------------------------------------------------
// Strided 3x3 pad-1 convolution, INPUT GRADIENT, all stride-parity classes in ONE launch on halo-staged tiles
// (resnet18.py:26,59-63: conv1 of each stage's first block, stride (2,1) / (2,2), and its 1x1 downsample branch).
//
// dx[hi][wi] receives the taps (dy, dx) with (hi + 1 - dy) % 2 == 0 and (wi + 1 - dx) % sw == 0, i.e. a pixel of
// parity class (a, b) = (hi & 1, wi % sw) contracts a fixed tap subset against dY pixels at FIXED offsets from its
// coarse position (hq, wq) = (hi >> 1, wi / sw):
//     a = 0: kernel row 1 <- dY row hq          a = 1: kernel row 0 <- dY row hq + 1, kernel row 2 <- dY row hq
//     sw = 1:        kernel columns 0, 1, 2 <- dY columns wq + 1, wq, wq - 1
//     sw = 2, b = 0: kernel column 1 <- wq      b = 1: kernel column 0 <- wq + 1, kernel column 2 <- wq
// An M tile is 256 consecutive coarse pixels of one coarse row and ONE class: per (kernel row, 64-channel chunk) ONE
// halo tile of dY (coarse pixels wq0 - 1 .. wq0 + 256) serves the class's 1-3 column taps, exactly as in the stride-1
// kernel above -- the per-class launches of the generic gather kernel staged one 256 x 64 A tile per tap, every class
// launch re-read dY, and the class rows were scattered by a per-row index remap.  Tiles of the classes of one coarse
// row are neighbours in the grid (same XCD: dY rows are read once), the heavier classes first.
// A2 (HtrvtGemmDesc.A2): the gradient of the block's 1x1 downsample conv output rides along as three more single-tap
// groups of the class-(0, 0) tiles (source = A + extra_off, weight slot 9), as in the per-class form.
// Groups have 1, 2 or 3 k-tiles; the next group's halo tile is issued during the current group's first k-tile(s) and
// always BEFORE the B pieces of the group's last k-tile, so the counted wait of that k-tile covers it.
// ---------------------------------------------------------------------------------------------
template <int BN, class P>
__device__ __forceinline__ void gemm_halo_s2_body(const P& p, const int block_x) {
  using H = HaloGeo<BN>;
  constexpr int BM = 256, NWC = 8, NW_TOTAL = 12, TM = 2, TN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int ntiles = p.tiles_m * p.tiles_n;
  int id = block_x;
  if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int n0 = tile_n * BN;
  // tile_m -> (coarse row, class, segment): classes of one coarse row are neighbours, heaviest first
  const int sw = p.sw, ncls = 2 * sw;
  const int segs = p.Wq >> 8;                               // 256-pixel segments per coarse row (host: Wq % 256 == 0)
  const int per_row = ncls * segs;
  const int rowq = tile_m / per_row, rem = tile_m - rowq * per_row;
  const int cidx = rem / segs, seg = rem - cidx * segs;
  const int ca = 1 - cidx / sw;                             // row parity of this tile's pixels
  const int cb = sw == 2 ? 1 - (cidx & 1) : 0;              // column parity (sw = 2)
  const int bimg = rowq / p.Hq, hq = rowq - bimg * p.Hq;
  const int wq0 = seg << 8;
  const int Hh = p.Ho, Ww = p.Wo, Cs = p.Co;                // gathered tensor dY [B, Ho, Wo, Co]: Ho = Hq, Wo = Wq
  // kernel rows / columns of this class, 2 bits per entry (wave-uniform scalars)
  const int nrow = ca ? 2 : 1;
  const int dyp = ca ? (0 | (2 << 2)) : 1;                  // kernel row of row slot r
  const int dhp = ca ? (1 | (0 << 2)) : 0;                  // dY row = hq + dh
  const int ncol = sw == 1 ? 3 : (cb ? 2 : 1);
  const int dxp = sw == 1 ? (0 | (1 << 2) | (2 << 4)) : (cb ? (0 | (2 << 2)) : 1);   // kernel column of column slot j
  const int shp = sw == 1 ? (2 | (1 << 2) | (0 << 4)) : (cb ? (2 | (1 << 2)) : 1);   // halo row shift of column slot j
  const int NC = p.Cpad / BK;
  const bool has_x = p.extra_off != 0 && ca == 0 && cb == 0;
  const int nmain = nrow * NC;                              // groups of the 3x3 taps
  const int ngroups = nmain + (has_x ? NC : 0);
  const int xtap = p.kh * p.kw;                             // weight slot of the 1x1 downsample conv
  const int kend = (xtap + 1) * p.Cpad;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < NWC;
  const int lw = (wave - NWC) & 3;
  const int wm = (wave >> 1) & 3, wn = wave & 1;
  if (!consumer) __builtin_amdgcn_s_setprio(3);

  DmaLoader<BN, HTRVT_KMAJOR, 0, 4> lb;
  lb.init(p, p.B, p.ldb, n0, p.N, lw, lane);
  const unsigned long long ba = (unsigned long long)p.A;
  const i32x4_t rsrcA = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
  const int rho0 = lw * 8 + (lane >> 3);
  const int cgA = (lane & 7) ^ Geo<BM>::swz(rho0);
  const unsigned lane_off = (unsigned)((wq0 - 1 + rho0) * Cs + cgA * 8) * 2u;
  const unsigned lds0 = lds_addr_of(smem);

  // Row slots of this tile: slot s < nrow = a kernel row of the 3x3 (dY row hq + dh), slot nrow = the 1x1 downsample
  // gradient (A2).  abase[s]: byte offset of (bimg, source row, pixel 0, channel 0) from p.A, OOB for a row outside dY.
  unsigned abase0, abase1, abasex;
  {
    const int h0 = hq + (dhp & 3), h1 = hq + ((dhp >> 2) & 3);
    abase0 = (unsigned)h0 < (unsigned)Hh ? (unsigned)(((bimg * Hh + h0) * Ww) * Cs) * 2u : OOB;
    abase1 = (nrow > 1 && (unsigned)h1 < (unsigned)Hh) ? (unsigned)(((bimg * Hh + h1) * Ww) * Cs) * 2u : OOB;
    abasex = has_x ? (unsigned)(((bimg * Hh + hq) * Ww) * Cs) * 2u + p.extra_off : OOB;
  }
  const int nslot = nrow + (has_x ? 1 : 0);
  // byte offset of (slot as, chunk ac)'s first pixel row, OOB for a row outside dY / past the last group.  (Formed HERE, on
  // plain values: inside a by-reference lambda hipcc turned the select over the three captured bases into a run-time
  // index into the closure object, which then lived in scratch memory together with the DMA descriptors.)
  auto group_base = [](int as, int ac, int nslot_, int nrow_, unsigned b0, unsigned b1, unsigned bx) -> unsigned {
    const unsigned rb = as >= nslot_ ? OOB : (as == nrow_ ? bx : (as == 0 ? b0 : b1));
    return rb < OOB ? rb + (unsigned)(ac * BK) * 2u : OOB;
  };
  // the halo tile at `gbase` (chunk ac) -> A stage `ast`, half `half`; gbase = OOB: zero fill
  auto issueA = [&](int half, int ast, unsigned gbase, int ac) {
    const bool chok = gbase < OOB && ac * BK + cgA * 8 < Cs;
#pragma unroll
    for (int i = 0; i < H::NP_AH; ++i) {
      const int ii = half * H::NP_AH + i;
      const int rho = rho0 + 32 * ii;
      const int w = wq0 - 1 + rho;
      const bool v = chok && rho < 258 && (unsigned)w < (unsigned)Ww;
      const unsigned voff = v ? gbase + lane_off + (unsigned)(32 * ii * Cs) * 2u : OOB;
      dma16(rsrcA, __builtin_amdgcn_readfirstlane(lds0 + ast * H::A_STAGE + (lw + 4 * ii) * 1024), voff);
    }
  };
  // the k-tile sequence as the B loader walks it (two k-tiles ahead of the multiply): (slot, chunk, column slot)
  int bs = 0, bc = 0, bj = 0;
  auto issueB = [&](int bst) {
    int k0 = kend;                                          // past the last k-tile: zero fill
    if (bs < nslot) {
      const bool x = bs == nrow;
      const int tap = x ? xtap : ((dyp >> (2 * bs)) & 3) * 3 + ((dxp >> (2 * bj)) & 3);
      k0 = tap * p.Cpad + bc * BK;
      if (++bj == (x ? 1 : ncol)) {
        bj = 0;
        if (++bc == NC) {
          bc = 0;
          ++bs;
        }
      }
    }
    lb.template issue<true>(p, lds0 + H::B_BASE + bst * H::B_STAGE, k0, kend, lw);
  };

  // C rows: pixel (bimg, 2 hq + ca, sw (wq0 + r) + cb) of the NHWC gradient, r = 0 .. 255
  const int m0 = (bimg * p.Hi + 2 * hq + ca) * p.Wi + sw * wq0 + cb;

  // Consumer and loader waves run SEPARATE copies of the loop (same barriers): straight-line fragment reads + MFMAs between
  // barriers on one side, nothing but address arithmetic and DMA issue on the other (gemm_halo_body: -4 ... -8 % per launch
  // against the one-loop form in which every wave walked both sides' branches).
  if (consumer) {
    f32x4h_t a4[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) a4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    int bst = 0;
    for (int g = 0; g < ngroups; ++g) {
      const int n = g < nmain ? ncol : 1;
      const char* sa = smem + (g & 1) * H::A_STAGE;
      for (int j = 0; j < n; ++j) {
        const int shift = g < nmain ? (shp >> (2 * j)) & 3 : 1;
        halo_mma16<BN, TM, TN>(a4, sa, smem + H::B_BASE + bst * H::B_STAGE, shift, wm, wn, lane);
        __builtin_amdgcn_s_barrier();
        bst = bst == 2 ? 0 : bst + 1;
      }
    }
    __builtin_amdgcn_s_barrier();
    f32x16_t acc[TM][TN];
    halo_acc16_to_32<TM, TN>(acc, a4);
    epilogue_staged<TN, BN, BM, NW_TOTAL, true, false, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, true, sw);
  } else {
    // ---- prologue: halo tile of group 0, B of k-tiles 0 and 1 ----
    const unsigned gb0 = group_base(0, 0, nslot, nrow, abase0, abase1, abasex);
    issueA(0, 0, gb0, 0);
    issueA(1, 0, gb0, 0);
    issueB(0);
    issueB(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
    __builtin_amdgcn_s_barrier();
    int bst = 0;                     // B stage of the k-tile being multiplied; the loaders fill (bst + 2) % 3
    int gs = 0, gc = 0;              // (slot, chunk) of group g
    for (int g = 0; g < ngroups; ++g) {
      const int n = g < nmain ? ncol : 1;
      int ns = gs, nc = gc + 1;      // group g + 1
      if (nc == NC) {
        nc = 0;
        ++ns;
      }
      const unsigned ngb = group_base(ns, nc, nslot, nrow, abase0, abase1, abasex);
      const int nast = (g + 1) & 1;
      for (int j = 0; j < n; ++j) {
        const bool last = j == n - 1;
        const int fill = bst >= 1 ? bst - 1 : 2;          // (bst + 2) % 3
        if (n == 1) {             // both halves of the next halo tile, then B: the wait below covers the halo tile
          issueA(0, nast, ngb, nc);
          issueA(1, nast, ngb, nc);
          issueB(fill);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
        } else if (n == 2) {      // k-tile 0: B + both halves (they have the group's second k-tile to land); k-tile 1: B
          issueB(fill);
          if (j == 0) {
            issueA(0, nast, ngb, nc);
            issueA(1, nast, ngb, nc);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + 2 * H::NP_AH) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
          }
        } else {                  // three k-tiles: the schedule of the stride-1 kernel
          issueB(fill);
          if (!last) {
            issueA(j, nast, ngb, nc);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NP_AH) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
          }
        }
        __builtin_amdgcn_s_barrier();
        bst = bst == 2 ? 0 : bst + 1;
      }
      gs = ns;
      gc = nc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill pieces issued past the last k-tile
    __builtin_amdgcn_s_barrier();
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    epilogue_staged<TN, BN, BM, NW_TOTAL, true, false, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, false, sw);
  }
}

template <int BN>
__global__ __launch_bounds__(768) void gemm_halo_s2_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_halo_s2_body<BN>(*kp, (int)blockIdx.x);
}

template <int BN>
int launch_halo_s2(const KParams& p, hipStream_t st) {
  constexpr int LDS = HaloGeo<BN>::LDS_BYTES;
  static bool attr_done = false;
  auto kern = gemm_halo_s2_kernel<BN>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(768), LDS, st, p);
  set_last_kernel("gemm_halo_s2_kernel<%d>", BN);
  const int rc = check_launch("gemm_halo_s2_kernel");
  return rc ? rc : 1;
}

// ---------------------------------------------------------------------------------------------
// 3x3 pad-1 convolution FORWARD with a COLUMN stride of 2 on halo-staged tiles (resnet18.py:26: conv1 of layer2.0 / layer3.0,
// stride (2,2); round 5 -- these launches ran on the generic per-row gather until now).
//
// Output pixel (ho, wo) reads x(ho sh + dy - 1, 2 wo + dx - 1).  An M tile is 256 consecutive output pixels wo0 .. wo0 + 255 of
// one output row: its column taps dx = 0 and dx = 2 read the ODD input pixels 2 wo0 - 1, 2 wo0 + 1, ... (dx = 2 the same
// pixels as dx = 0, one tile row further down) and dx = 1 the EVEN pixels 2 wo0, 2 wo0 + 2, ...  Per (kernel row, 64-channel
// chunk) = "unit": ONE odd image of 257 pixels serves two k-tiles (row shift 0 / 1) and one even image of 256 pixels the third
// -- 36 + 32 DMA pieces instead of 3 x 32, and the loader waves form one address per piece instead of decoding a pixel per
// row and tap.  k-tile order inside a unit: dx = 0, dx = 2 (odd image, LDS stage X), dx = 1 (even image, stage Y).  B runs
// two k-tiles ahead through three stages as in gemm_halo_body; the even image of a unit is issued during the unit's first
// k-tile and has two k-tiles to land; the odd image of the NEXT unit can only be issued when the current unit's second
// k-tile has read stage X for the last time, i.e. during the third k-tile, and must land within it (a second odd stage
// would need 170 KB with three B stages).
// ---------------------------------------------------------------------------------------------
template <int BN>
struct HaloFs2Geo {
  static constexpr int NPO = 9, NPE = 8;                  // 1-KiB pieces per loader wave: odd image (288 rows, 257 used), even image (256 rows)
  static constexpr int X_BYTES = 4 * NPO * 1024, Y_BYTES = 4 * NPE * 1024;
  static constexpr int B_STAGE = Geo<BN>::BYTES;
  static constexpr int NP_B = B_STAGE / 1024 / 4;
  static constexpr unsigned Y_BASE = X_BYTES, B_BASE = X_BYTES + Y_BYTES;
  static constexpr int LDS_BYTES = X_BYTES + Y_BYTES + 3 * B_STAGE;
  static_assert(B_STAGE % 4096 == 0 && LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(LDS_BYTES >= BN * Stg<256>::CST, "the staged epilogue's column-major image");
};

template <int BN, bool F32, class P>
__device__ __forceinline__ void gemm_halo_fs2_body(const P& p, const int block_x) {
  using H = HaloFs2Geo<BN>;
  constexpr int BM = 256, NWC = 8, NW_TOTAL = 12, TM = 2, TN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int ntiles = p.tiles_m * p.tiles_n;
  int id = block_x;
  if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int rowo = m0 / p.Wo, wo0 = m0 - rowo * p.Wo;     // output row (b * Ho + ho), first output column of this tile
  const int bimg = rowo / p.Ho, ho = rowo - bimg * p.Ho;
  const int NC = p.Cpad / BK;
  const int NU = 3 * NC;                                  // units = (kernel row, chunk); k-tiles = 3 NU

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave >> 1) & 3, wn = wave & 1;

  if (wave < NWC) {
    // ================================ consumer waves ================================
    f32x4h_t a4[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) a4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    for (int u = 0; u < NU; ++u) {
      halo_mma16<BN, TM, TN>(a4, smem, smem + H::B_BASE + 0 * H::B_STAGE, 0, wm, wn, lane);               // dx = 0
      __builtin_amdgcn_s_barrier();
      halo_mma16<BN, TM, TN>(a4, smem, smem + H::B_BASE + 1 * H::B_STAGE, 1, wm, wn, lane);               // dx = 2
      __builtin_amdgcn_s_barrier();
      halo_mma16<BN, TM, TN>(a4, smem + H::Y_BASE, smem + H::B_BASE + 2 * H::B_STAGE, 0, wm, wn, lane);   // dx = 1
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (F32) {
      halo_epilogue_f32<BN, TM, TN, NW_TOTAL * 64, true>(a4, p, m0, n0, wm, wn, tile_m, lane, smem, true);
    } else {
      f32x16_t acc[TM][TN];
      halo_acc16_to_32<TM, TN>(acc, a4);
      epilogue_staged<TN, BN, BM, NW_TOTAL, false, true, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, true);
    }
  } else {
    // ================================ loader waves ================================
    const int lw = (wave - NWC) & 3;
    __builtin_amdgcn_s_setprio(3);
    DmaLoader<BN, HTRVT_KMAJOR, 0, 4> lb;
    lb.init(p, p.B, p.ldb, n0, p.N, lw, lane);
    const unsigned long long ba = (unsigned long long)p.A;
    const i32x4_t rsrcA = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
    const int Hh = p.Hi, Ww = p.Wi, Cs = p.Ci;
    const int rho0 = lw * 8 + (lane >> 3);                // this lane's image row in piece lw; piece lw + 4 i is 32 i rows further down
    const int cgA = (lane & 7) ^ Geo<BM>::swz(rho0);
    const int wsrc0 = 2 * (wo0 + rho0) - 1;               // source pixel of odd-image row rho0 (even image: + 1); 32 image rows = 64 pixels
    const unsigned lane_off = (unsigned)(wsrc0 * Cs + cgA * 8) * 2u;   // may wrap for wsrc0 < 0: masked below
    const unsigned lds0 = lds_addr_of(smem);
    // image `par` (0 odd -> stage X, 1 even -> stage Y) of unit (udy, ucc); uvalid false: zero fill
    auto issueA = [&](int par, int udy, int ucc, bool uvalid) {
      const int hh = ho * p.sh + udy - 1;
      const bool rowok = uvalid && (unsigned)hh < (unsigned)Hh;
      const unsigned gbase = (unsigned)(((bimg * Hh + hh) * Ww + par) * Cs + ucc * BK) * 2u;
      const bool chok = ucc * BK + cgA * 8 < Cs;
      if (par == 0) {
#pragma unroll
        for (int i = 0; i < H::NPO; ++i) {
          const int rho = rho0 + 32 * i, w = wsrc0 + 64 * i;
          const bool v = rowok && chok && rho < 257 && (unsigned)w < (unsigned)Ww;
          dma16(rsrcA, __builtin_amdgcn_readfirstlane(lds0 + (lw + 4 * i) * 1024), v ? gbase + lane_off + (unsigned)(64 * i * Cs) * 2u : OOB);
        }
      } else {
#pragma unroll
        for (int i = 0; i < H::NPE; ++i) {
          const int w = wsrc0 + 1 + 64 * i;
          const bool v = rowok && chok && (unsigned)w < (unsigned)Ww;
          dma16(rsrcA, __builtin_amdgcn_readfirstlane(lds0 + H::Y_BASE + (lw + 4 * i) * 1024), v ? gbase + lane_off + (unsigned)(64 * i * Cs) * 2u : OOB);
        }
      }
    };
    auto issueB = [&](int bst, int udy, int ucc, int dx, bool uvalid) {
      lb.template issue<true>(p, lds0 + H::B_BASE + bst * H::B_STAGE, uvalid ? (udy * 3 + dx) * p.Cpad + ucc * BK : p.K, p.K, lw);
    };
    // ---- prologue: odd image of unit 0, B of k-tiles 0 (dx = 0) and 1 (dx = 2) ----
    issueA(0, 0, 0, true);
    issueB(0, 0, 0, 0, true);
    issueB(1, 0, 0, 2, true);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
    __builtin_amdgcn_s_barrier();
    int udy = 0, ucc = 0;
    for (int u = 0; u < NU; ++u) {
      int ndy = udy, ncc = ucc + 1;
      if (ncc == NC) {
        ncc = 0;
        ++ndy;
      }
      const bool nvalid = u + 1 < NU;
      // k-tile 3u (dx = 0): B of this unit's third k-tile; its even image (two k-tiles to land)
      issueB(2, udy, ucc, 1, true);
      issueA(1, udy, ucc, true);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NPE) : "memory");
      __builtin_amdgcn_s_barrier();
      // k-tile 3u + 1 (dx = 2): B of the next unit's first k-tile; the even image has landed behind this wait
      issueB(0, ndy, ncc, 0, nvalid);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
      __builtin_amdgcn_s_barrier();
      // k-tile 3u + 2 (dx = 1): stage X is free -- the next unit's odd image, then B of its second k-tile
      issueA(0, ndy, ncc, nvalid);
      issueB(1, ndy, ncc, 2, nvalid);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
      __builtin_amdgcn_s_barrier();
      udy = ndy;
      ucc = ncc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill pieces issued past the last k-tile
    __builtin_amdgcn_s_barrier();
    if constexpr (F32) {
      f32x4h_t z4[2 * TM][2 * TN];
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) z4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
      halo_epilogue_f32<BN, TM, TN, NW_TOTAL * 64, true>(z4, p, m0, n0, wm, wn, tile_m, lane, smem, false);
    } else {
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    epilogue_staged<TN, BN, BM, NW_TOTAL, false, true, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, false);
    }
  }
}

template <int BN, bool F32 = false>
__global__ __launch_bounds__(768) void gemm_halo_fs2_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_halo_fs2_body<BN, F32>(*kp, (int)blockIdx.x);
}

template <int BN, bool F32 = false>
int launch_halo_fs2(const KParams& p, hipStream_t st) {
  constexpr int LDS = HaloFs2Geo<BN>::LDS_BYTES;
  static bool attr_done = false;
  auto kern = gemm_halo_fs2_kernel<BN, F32>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(768), LDS, st, p);
  if (F32) set_last_kernel("gemm_halo_fs2_kernel<%d, f32>", BN);
  else set_last_kernel("gemm_halo_fs2_kernel<%d>", BN);
  const int rc = check_launch("gemm_halo_fs2_kernel");
  return rc ? rc : 1;
}

// ---------------------------------------------------------------------------------------------
// gemm_halo_body, round 5: consumer and loader waves run SEPARATE copies of the loop (same barriers) -- straight-line fragment
// reads + MFMAs between barriers on one side, nothing but address arithmetic and DMA issue on the other.  In the one-loop form
// above every wave walked both sides' branches and the register allocation, the scalar state and the instruction stream of
// either role carried the other's: same-box A/B -5 ... -9 % per launch on every stride-1 3x3 convolution of the stem
// (profiles/r05_experiments.md).  The epilogue is inlined once per role (the loader copy with a dead accumulator set).
// ---------------------------------------------------------------------------------------------
// (DGRAD / forward and the row stride as in gemm_halo_body_oneloop above)
template <int BN, bool DGRAD, bool F32, class P>
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
  const int rowi = m0 / Ww, w0 = m0 - rowi * Ww;
  const int Hm = DGRAD ? Hh : p.Ho;                     // forward with a row stride: M rows are OUTPUT rows
  const int bimg = rowi / Hm, hrow = rowi - bimg * Hm;
  const int NC = p.Cpad / BK;
  const int NG = 3 * NC;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave >> 1) & 3, wn = wave & 1;

  if (wave < NWC) {
    // ================================ consumer waves: fragments + MFMA, nothing else ================================
    HTRVT_STAMP(0);
    f32x4h_t a4[2 * TM][2 * TN];
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) a4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    HTRVT_STAMP(1);
    HTRVT_STAMP(2);
    for (int g = 0; g < NG; ++g) {
      const char* sa = smem + (g & 1) * H::A_STAGE;
      halo_mma16<BN, TM, TN>(a4, sa, smem + H::B_BASE + 0 * H::B_STAGE, DGRAD ? 2 : 0, wm, wn, lane);
      __builtin_amdgcn_s_barrier();
      halo_mma16<BN, TM, TN>(a4, sa, smem + H::B_BASE + 1 * H::B_STAGE, 1, wm, wn, lane);
      __builtin_amdgcn_s_barrier();
      halo_mma16<BN, TM, TN>(a4, sa, smem + H::B_BASE + 2 * H::B_STAGE, DGRAD ? 0 : 2, wm, wn, lane);
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();
    HTRVT_STAMP(3);
    if constexpr (F32) {
      halo_epilogue_f32<BN, TM, TN, NW_TOTAL * 64, !DGRAD>(a4, p, m0, n0, wm, wn, tile_m, lane, smem, true);
    } else {
      f32x16_t acc[TM][TN];
      halo_acc16_to_32<TM, TN>(acc, a4);
      epilogue_staged<TN, BN, BM, NW_TOTAL, DGRAD, !DGRAD, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, true);
    }
    HTRVT_STAMP(6);
#ifdef HTRVT_EXP_STAMP      // experiment builds: when have this wave's stores drained, and on which CU did the tile run
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HTRVT_STAMP(7);
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      htrvt_dbg[blockIdx.x * 16 + 8] = hw;
    }
#endif
  } else {
    // ================================ loader waves: LDS-DMA of the operands, then the side tile ================================
    const int lw = (wave - NWC) & 3;
    __builtin_amdgcn_s_setprio(3);
    DmaLoader<BN, HTRVT_KMAJOR, 0, 4> lb;
    lb.init(p, p.B, p.ldb, n0, p.N, lw, lane);
    const unsigned long long ba = (unsigned long long)p.A;
    const i32x4_t rsrcA = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
    const int rho0 = lw * 8 + (lane >> 3);
    const int cgA = (lane & 7) ^ Geo<BM>::swz(rho0);
    const unsigned lane_off = (unsigned)((w0 - 1 + rho0) * Cs + cgA * 8) * 2u;
    const unsigned lds0 = lds_addr_of(smem);
    auto issueA = [&](int half, int ast, int gdy, int gcc) {
      const int hh = DGRAD ? hrow + 1 - gdy : hrow * p.sh + gdy - 1;
      const bool rowok = (unsigned)hh < (unsigned)Hh;
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
    auto issueB = [&](int bst, int gdy, int gcc, int dx) {
      lb.template issue<true>(p, lds0 + H::B_BASE + bst * H::B_STAGE, (gdy * 3 + dx) * p.Cpad + gcc * BK, p.K, lw);
    };
    // ---- prologue: halo tile of group 0, B of k-tiles 0 and 1 ----
    issueA(0, 0, 0, 0);
    issueA(1, 0, 0, 0);
    issueB(0, 0, 0, 0);
    issueB(1, 0, 0, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
    __builtin_amdgcn_s_barrier();
    int gdy = 0, gcc = 0;
    for (int g = 0; g < NG - 1; ++g) {       // every group but the last: the schedule of gemm_halo_body
      int ndy = gdy, ncc = gcc + 1;
      if (ncc == NC) {
        ncc = 0;
        ++ndy;
      }
      const int nast = (g + 1) & 1;
      issueB(2, gdy, gcc, 2);
      issueA(0, nast, ndy, ncc);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NP_AH) : "memory");
      __builtin_amdgcn_s_barrier();
      issueB(0, ndy, ncc, 0);
      issueA(1, nast, ndy, ncc);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B + H::NP_AH) : "memory");
      __builtin_amdgcn_s_barrier();
      issueB(1, ndy, ncc, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
      __builtin_amdgcn_s_barrier();
      gdy = ndy;
      gcc = ncc;
    }
    // ---- last group: its third B tile, nothing left to stage for a next group ----
    issueB(2, gdy, gcc, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NP_B) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    if constexpr (F32) {
      f32x4h_t z4[2 * TM][2 * TN];
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) z4[i][j] = f32x4h_t{0.f, 0.f, 0.f, 0.f};
      halo_epilogue_f32<BN, TM, TN, NW_TOTAL * 64, !DGRAD>(z4, p, m0, n0, wm, wn, tile_m, lane, smem, false);
    } else {
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    epilogue_staged<TN, BN, BM, NW_TOTAL, DGRAD, !DGRAD, P, 16>(acc, p, 0ll, m0, n0, wm, wn, tile_m, lane, wave, smem, false);
    }
  }
}

template <int BN, bool DGRAD, bool F32 = false>
__global__ __launch_bounds__(768) void gemm_halo_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef HTRVT_HALO_ONELOOP      // A/B builds: consumer and loader waves in ONE copy of the loop (rounds 3-4)
  gemm_halo_body_oneloop<BN, DGRAD>(*kp, (int)blockIdx.x);
#else
  gemm_halo_body<BN, DGRAD, F32>(*kp, (int)blockIdx.x);
#endif
}

template <int BN, bool DGRAD, bool F32 = false>
int launch_halo(const KParams& p, hipStream_t st) {
  using H = HaloGeo<BN>;
  static bool attr_done = false;
  auto kern = gemm_halo_kernel<BN, DGRAD, F32>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, H::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", H::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(768), H::LDS_BYTES, st, p);
  if (F32) set_last_kernel("gemm_halo_kernel<%d, %s, f32>", BN, DGRAD ? "true" : "false");
  else set_last_kernel("gemm_halo_kernel<%d, %s>", BN, DGRAD ? "true" : "false");
  const int rc = check_launch("gemm_halo_kernel");
  return rc ? rc : 1;
}

}  // namespace
