// gemm8p_impl.h -- bfloat16 GEMM / implicit-GEMM convolution on the "8-phase" schedule (round 3).
//
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)   -- same contract as htrvt_gemm (include/htrvt.h)
//
// Replaces, for the shapes it serves, the one-barrier-per-k-tile LDS-DMA kernel of gemm_dma_impl.h: that loop pays
// one DMA round trip per k-tile (two LDS stages, the next DMA cannot start before the barrier behind the previous
// one's arrival) and its LDS-staged epilogue does not overlap with anything.  Here:
//
//  * 256 x BN x 64 tiles (BN = 256: 2 x 4 waves of 128 x 64; BN = 192: 4 x 2 waves of 64 x 96), 8 waves = two groups
//    of four that share the 4 SIMDs pairwise and run ONE BARRIER APART: while a group multiplies (12-16 MFMA
//    16x16x32 between two barriers) its SIMD partners read their fragments from LDS and issue DMA, then the roles swap,
//    so the matrix pipe of every SIMD always has a wave on it.
//  * operands arrive by LDS-DMA in HALF tiles (A rows 0-127 / 128-255, B columns likewise): a k-tile is four phases,
//    each phase reads one half tile's fragments, issues the DMA of ONE half tile of a later k-tile into a half that
//    was consumed two phases earlier, and multiplies one quadrant of the wave's block.  Three half tiles are always in
//    flight; the only wait is a counted `s_waitcnt vmcnt(6)` once per k-tile (never 0 inside the loop).
//  * the product is computed "n on the rows": MFMA a-operand = B (weight) rows, b-operand = A (activation) rows, so a
//    lane's four accumulator registers are four CONSECUTIVE OUTPUT COLUMNS of one row; the B rows are loaded into LDS
//    in a permuted order such that the lane's registers of two (three) neighbouring column tiles are 8 (12) consecutive
//    columns -> the epilogue stores 16-byte row pieces straight from the accumulators (bias, GELU, residual, ReLU mask,
//    BatchNorm sums applied in registers): no LDS staging, no barrier, the LDS is free for the next DMA.
//
// LDS image of a K-major half tile: [rows][128 B], 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (applied to
// the DMA's per-lane SOURCE chunk and again on the read): conflict-free ds_read_b128 for the 16x16x32 operand map.
#pragma once
#include <type_traits>

#include "gemm_common.h"

using namespace htrvt;

namespace g8 {

constexpr int BK = 64;
constexpr unsigned OOB = 0x80000000u;

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 B -> LDS [lds_addr, lds_addr + 1 KiB).  Inline asm: hipcc must not know that this
// writes LDS (it would order every later ds_read behind it with s_waitcnt vmcnt(0)); the kernel counts the pieces itself.
__device__ __forceinline__ void dma16(const i32x4_t& rsrc, unsigned lds_addr, unsigned voff) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_addr), "s"(rsrc)
      : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ i32x4_t make_rsrc(const char* base) {
  const unsigned long long ba = (unsigned long long)base;  // raw buffer, stride 0, 2 GiB of records
  return i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
}

template <int BN_, int WARPS_M_, int WARPS_N_>
struct Cfg {
  static constexpr int BM = 256, BN = BN_, WARPS_M = WARPS_M_, WARPS_N = WARPS_N_;
  static_assert(WARPS_M * WARPS_N == 8, "8 waves");
  static constexpr int HM = BM / 2, HN = BN / 2;            // rows of an A / B half tile
  static constexpr int SM = HM / WARPS_M, SN = HN / WARPS_N;  // a wave's rows / columns inside one half tile
  static constexpr int MT = SM / 16, NT = SN / 16;          // 16 x 16 MFMA tiles per quadrant
  static_assert(SM % 16 == 0 && SN % 16 == 0 && (NT == 2 || NT == 3), "quadrant shape");
  static constexpr int A_HALF = HM * 128, B_HALF = HN * 128;   // bytes
  static constexpr int BUF = 2 * A_HALF + 2 * B_HALF;          // one k-tile
  static constexpr int A_PIECES = HM / 8, B_PIECES = HN / 8;   // 1-KiB DMA pieces per half tile
  static constexpr int NPW = 2;                                // pieces per wave and half tile (B: 12 of 16 real when HN = 96)
  static_assert(A_PIECES == 16 && B_PIECES <= 16, "two pieces per wave");
  static constexpr int SCRATCH = 2 * BUF;                      // 1 KiB target of the dummy pieces
  static constexpr int LDS_BYTES = 2 * BUF + 1024;
  static constexpr int MFMA_PER_PHASE = MT * NT * 2;
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// LDS row r (0 .. HN-1) of a B half tile holds column `bcol(r)` of that half: the wave-column block wc owns LDS rows
// [wc*SN, wc*SN + SN); inside it MFMA tile nt, operand row i (= accumulator row group i>>2, register i&3) is column
// 4*NT*(i>>2) + 4*nt + (i&3): lane group g of the accumulators then holds columns 4*NT*g .. 4*NT*g + 4*NT-1 contiguously.
template <class C>
__device__ __forceinline__ int bcol(int r) {
  const int wc = r / C::SN, rr = r - wc * C::SN;
  const int nt = rr >> 4, i = rr & 15;
  return wc * C::SN + 4 * C::NT * (i >> 2) + 4 * nt + (i & 3);
}

// ---------------------------------------------------------------------------------------------
// DMA source offsets.  Every thread owns, per half tile, NPW pieces = NPW (row, chunk) pairs; the k position of a
// k-tile is added when the piece is issued.  An invalid row / column keeps an offset >= 2^31 (delivers zeros).
// GATHER: 0 plain rows, 1 conv-forward rows (output pixels), 2 conv-dgrad rows (input pixels, optionally one parity class)
// ---------------------------------------------------------------------------------------------
template <class C, int GATHER>
struct ALoader {
  unsigned off0[2][C::NPW];     // plain: row byte offset + chunk; gather: image base byte offset (or OOB)
  int hw[2][C::NPW];            // gather: (c0 << 16) | (c1 & 0xffff): window origin of the row's pixel
  unsigned tapoff[2][C::NPW];   // gather: byte offset of (pixel reached through the current tap) + chunk, or OOB
  int ck;                       // element offset of this lane's chunk inside a k-tile (after the swizzle)
  int cur_ti;
  int k0, ti, cbase;            // the k-tile the next half tile belongs to: first k, and (gathers) its tap position / channel base
  i32x4_t rsrc;

  template <class P>
  __device__ __forceinline__ void init(const P& p, const char* base, int m0, int kbeg, int wave, int lane) {
    rsrc = make_rsrc(base);
    cur_ti = -1;
    k0 = kbeg;
    ti = 0;
    cbase = 0;
    if constexpr (GATHER != 0) {
      ti = kbeg / p.Cpad;
      cbase = kbeg - ti * p.Cpad;
    }
    const int rl = lane >> 3;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) {
        const int r = (wave + 8 * i) * 8 + rl;             // LDS row inside the half tile
        const int cg = (lane & 7) ^ swz(r);
        const int row = m0 + x * C::HM + r;
        ck = cg * 8;     // the same for every piece of a thread: r = 8 wave + 64 i + rl, so swz(r) = 4 (wave & 1) + (rl >> 1)
        const bool ok = row < p.M;
        if constexpr (GATHER == 0) {
          off0[x][i] = ok ? (unsigned)row * (unsigned)(p.lda * 2) + cg * 16 : OOB;
          hw[x][i] = 0;
        } else if constexpr (GATHER == 1) {
          const int hwn = p.Ho * p.Wo;
          const int b = row / hwn, rr = row - b * hwn;
          const int ho = rr / p.Wo, wo = rr - ho * p.Wo;
          off0[x][i] = ok ? (unsigned)b * (unsigned)(p.Hi * p.Wi * p.Ci * 2) + cg * 16 : OOB;
          hw[x][i] = ((ho * p.sh - p.ph) << 16) | ((wo * p.sw - p.pw) & 0xffff);
        } else {
          const int hwn = p.Hq * p.Wq;
          const int b = row / hwn, rr = row - b * hwn;
          int hi = rr / p.Wq, wi = rr - hi * p.Wq;
          if (p.cls_h >= 0) {
            hi = hi * p.sh + p.cls_h;
            wi = wi * p.sw + p.cls_w;
          }
          off0[x][i] = ok ? (unsigned)b * (unsigned)(p.Ho * p.Wo * p.Co * 2) + cg * 16 : OOB;
          hw[x][i] = ((hi + p.ph) << 16) | ((wi + p.pw) & 0xffff);
        }
        tapoff[x][i] = OOB;
      }
  }

  // half tile X of the current k-tile (half 0 is always issued before half 1; after half 1 the loader moves to the next
  // k-tile).  Past kend: zero fill.
  template <int X, class P>
  __device__ __forceinline__ void issue(const P& p, unsigned lds_half, int kend, int wave) {
    if constexpr (GATHER != 0 && X == 0) {
      if (ti != cur_ti && k0 < kend) {   // wave-uniform: the k loop enters a new tap
        cur_ti = ti;
        const int tap = (int)((p.tappack >> (4 * ti)) & 15ull);
        const int dy = tap / p.kw, dx = tap - dy * p.kw;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int i = 0; i < C::NPW; ++i) {
            const int c0 = hw[x][i] >> 16, c1 = (int)(short)(hw[x][i] & 0xffff);
            bool v;
            unsigned off = off0[x][i];
            if constexpr (GATHER == 1) {
              const int hi = c0 + dy, wi = c1 + dx;
              v = ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
              off += (unsigned)((hi * p.Wi + wi) * p.Ci) * 2;
            } else {
              const int th = c0 - dy, tw = c1 - dx;
              const int ho = th >> (p.sh - 1), wo = tw >> (p.sw - 1);
              v = (th >= 0) && (tw >= 0) && ((th & (p.sh - 1)) == 0) && ((tw & (p.sw - 1)) == 0) && (ho < p.Ho) && (wo < p.Wo);
              off += (unsigned)((ho * p.Wo + wo) * p.Co) * 2;
            }
            tapoff[x][i] = (v && off0[x][i] < OOB) ? off : OOB;
          }
      }
    }
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
      unsigned voff;
      if constexpr (GATHER == 0) {
        voff = (k0 + ck < kend) ? off0[X][i] + (unsigned)k0 * 2 : OOB;
      } else {
        const int cvalid = GATHER == 1 ? p.Ci : p.Co;
        voff = (k0 < kend && cbase + ck < cvalid) ? tapoff[X][i] + (unsigned)cbase * 2 : OOB;
      }
      dma16(rsrc, __builtin_amdgcn_readfirstlane(lds_half + (wave + 8 * i) * 1024), voff);
    }
    if constexpr (X == 1) {
      k0 += BK;
      if constexpr (GATHER != 0) {
        cbase += BK;
        if (cbase >= p.Cpad) {
          cbase -= p.Cpad;
          ++ti;
        }
      }
    }
  }
};

// B operand: plain K-major rows [N][ldb]; KMAP (conv forward / dgrad): k runs over a SELECTED tap list of the packed
// weights, column of the packed matrix = tap * Cpad + channel
template <class C, bool KMAP>
struct BLoader {
  unsigned off0[2][C::NPW];
  int ck;
  int k0, ti, cbase;
  i32x4_t rsrc;

  template <class P>
  __device__ __forceinline__ void init(const P& p, const char* base, int n0, int kbeg, int wave, int lane) {
    rsrc = make_rsrc(base);
    k0 = kbeg;
    ti = 0;
    cbase = 0;
    if constexpr (KMAP) {
      ti = kbeg / p.Cpad;
      cbase = kbeg - ti * p.Cpad;
    }
    const int rl = lane >> 3;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) {
        const int pi = wave + 8 * i;
        const int r = pi * 8 + rl;
        const int cg = (lane & 7) ^ swz(r);
        ck = cg * 8;                                         // the same for every piece of a thread (see ALoader)
        const int col = n0 + y * C::HN + bcol<C>(r < C::HN ? r : 0);
        const bool ok = pi < C::B_PIECES && col < p.N;       // pieces past the half tile (HN = 96) are dummies: zero fill into the scratch KiB
        off0[y][i] = ok ? (unsigned)col * (unsigned)(p.ldb * 2) + cg * 16 : OOB;
      }
  }

  template <int Y, class P>
  __device__ __forceinline__ void issue(const P& p, unsigned lds_half, unsigned lds_scratch, int kend, int wave) {
    int kcol = k0;
    if constexpr (KMAP) {
      const int tap = (int)((p.tappack >> (4 * (ti < 15 ? ti : 15))) & 15ull);
      kcol = cbase + tap * p.Cpad;
    }
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
      const unsigned voff = (k0 + ck < kend) ? off0[Y][i] + (unsigned)kcol * 2 : OOB;
      const bool dm = (C::B_PIECES < 16) && (wave + 8 * i >= C::B_PIECES);      // wave-uniform
      dma16(rsrc, __builtin_amdgcn_readfirstlane(dm ? lds_scratch : lds_half + (wave + 8 * i) * 1024), voff);
    }
    if constexpr (Y == 1) {
      k0 += BK;
      if constexpr (KMAP) {
        cbase += BK;
        if (cbase >= p.Cpad) {
          cbase -= p.Cpad;
          ++ti;
        }
      }
    }
  }
};

// one 16-row operand fragment of k-step s: lane (row l & 15, k chunk 4 s + (l >> 4)); `a0` = byte address for s = 0 of
// the fragment's first tile, the other k-step is the same address with bit 6 flipped (the swizzle is an XOR)
__device__ __forceinline__ bf16x8_t ldfrag(const char* lds, unsigned off) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + off));
}

// ---------------------------------------------------------------------------------------------
// Epilogue feature sets (compile time; the host picks the instantiation, anything else stays on the older kernels)
// ---------------------------------------------------------------------------------------------
constexpr int E_RES = 1;        // + residual (same shape / type as C)
constexpr int E_GELU = 2;       // exact-erf GELU, pre-activation saved to p.preact when non-null
constexpr int E_GELUGRAD = 4;   // multiply by GELU'(p.preact)
constexpr int E_CSTATS = 8;     // per-column sum / sum of squares of the accumulators -> p.colstats (conv forward)
constexpr int E_RELUMASK = 16;  // C = relu_src > 0 ? value : 0 (after the residual)
constexpr int E_BNB1 = 32;      // BatchNorm-backward sums against bnb_x[0]
constexpr int E_BNB2 = 64;      // ... and bnb_x[1]
constexpr int E_F32 = 128;      // float32 C, plain stores (slab or final), alpha / bias only
constexpr int E_SCALE_RELU = 256;  // eval-mode BatchNorm folded in: per-column scale (+ bias = shift) and ReLU last (act == 3)

template <class C, int GATHER, int EPI, class P>
__device__ __forceinline__ void gemm8p_body(const P& p, const int block_x) {
  constexpr int MT = C::MT, NT = C::NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> tile: consecutive ids share the A rows (all N tiles of one M tile), ids are dealt to the XCDs in
  //      contiguous chunks (blocks b and b + 8 share an XCD), bijective for any tile count ----
  const int ntiles = p.tiles_m * p.tiles_n;
  int id;
  {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = block_x & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (block_x >> 3);
  }
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

  const int z = blockIdx.z;
  const char* Ab = p.A;
  const char* Bb = p.B;
  long long coff = 0;
  {
    const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
    Ab += (zo * p.sA_o + zi * p.sA_i) * 2;
    Bb += (zo * p.sB_o + zi * p.sB_i) * 2;
    coff = zo * p.sC_o + zi * p.sC_i;
  }
  const int kbeg = 0, kend = p.K;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                                   // the two staggered groups: SIMD partners are w and w + 4
  const int wr = wave / C::WARPS_N, wc = wave - wr * C::WARPS_N;

  ALoader<C, GATHER> la;
  BLoader<C, GATHER != 0> lb;
  la.init(p, Ab, m0, kbeg, wave, lane);
  lb.init(p, Bb, n0, kbeg, wave, lane);

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

  // fragment read offsets (bytes inside a half tile), k-step 0; k-step 1 = ^ 64
  const int fr = lane & 15, fg = lane >> 4;
  const int ra = wr * C::SM + fr, rb = wc * C::SN + fr;
  const unsigned rdA = ra * 128 + ((fg ^ swz(ra)) << 4);
  const unsigned rdB = rb * 128 + ((fg ^ swz(rb)) << 4);

  // the loaders walk the k-tiles themselves: A half 0, A half 1, next k-tile ...; B likewise
  auto stageA = [&](auto xc, auto bufc) {
    constexpr int X = decltype(xc)::value, BUFI = decltype(bufc)::value;
    la.template issue<X>(p, lds0 + BUFI * C::BUF + (X ? OA1 : OA0), kend, wave);
  };
  auto stageB = [&](auto yc, auto bufc) {
    constexpr int Y = decltype(yc)::value, BUFI = decltype(bufc)::value;
    lb.template issue<Y>(p, lds0 + BUFI * C::BUF + (Y ? OB1 : OB0), lds0 + C::SCRATCH, kend, wave);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- prologue: k-tile 0 complete + the three half tiles of k-tile 1 the steady state has in flight at a k-tile's start ----
  stageB(I0{}, I0{});
  stageA(I0{}, I0{});
  stageB(I1{}, I0{});
  stageA(I1{}, I0{});
  stageB(I0{}, I1{});
  stageA(I0{}, I1{});
  stageB(I1{}, I1{});
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 from here on

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
      fa[i][0] = ldfrag(half, rdA + i * 2048);
      fa[i][1] = ldfrag(half, (rdA ^ 64) + i * 2048);
    }
  };
  auto readB = [&](const char* half, bf16x8_t (&f)[NT][2]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f[j][0] = ldfrag(half, rdB + j * 2048);
      f[j][1] = ldfrag(half, (rdB ^ 64) + j * 2048);
    }
  };
#define G8_MFMA_PHASE(X, Y, FB)                  \
  __builtin_amdgcn_s_barrier();                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_setprio(1);                 \
  mma(X, Y, FB);                                 \
  __builtin_amdgcn_s_setprio(0);                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();

  // one k-tile = four phases; BUFI: the LDS buffer it is multiplied from, kt its index
  auto ktile = [&](auto bufc) {
    constexpr int BUFI = decltype(bufc)::value;
    using BX = std::integral_constant<int, BUFI>;
    using BY = std::integral_constant<int, BUFI ^ 1>;
    const char* base = smem + BUFI * C::BUF;
    // phase 1: b0 then a0; DMA of A half 1 of k-tile kt+1 (other buffer; last read two phases ago)
    readB(base + OB0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    readA(base + OA0);
    __builtin_amdgcn_sched_barrier(0);
    stageA(I1{}, BY{});
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT * 2) : "memory");   // the b0 reads (issued first) have returned: B half 0 may be restaged next phase
    G8_MFMA_PHASE(I0{}, I0{}, fb0)
    // phase 2: b1; DMA of B half 0 of k-tile kt+2 (this buffer)
    readB(base + OB1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stageB(I0{}, BX{});
    G8_MFMA_PHASE(I0{}, I1{}, fb1)
    // phase 3: a1; DMA of A half 0 of k-tile kt+2
    readA(base + OA1);
    __builtin_amdgcn_sched_barrier(0);
    stageA(I0{}, BX{});
    G8_MFMA_PHASE(I1{}, I1{}, fb1)
    // phase 4: DMA of B half 1 of k-tile kt+2; everything older than the last three half tiles has landed = k-tile kt+1
    stageB(I1{}, BX{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    G8_MFMA_PHASE(I1{}, I0{}, fb0)
  };

  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    ktile(I0{});
    ktile(I1{});
  }
  if (kt < nkt) ktile(I0{});
#undef G8_MFMA_PHASE
  if (grp == 0) __builtin_amdgcn_s_barrier();   // group 0 joins group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill pieces issued past the last k-tile

  // ------------------------------------------------------------------ epilogue: straight from the accumulators
  // lane (g = lane >> 4, j = lane & 15): row m = m0 + x*128 + wr*SM + 16*i + j,
  //   columns n = n0 + y*HN + wc*SN + 4*NT*g + 4*t + r  (t = column tile, r = register) -> 4*NT consecutive columns
  // Straight-line code: every load and store is a buffer access whose offset is 2^31 (reads zeros / is dropped) for rows
  // >= M and columns >= N.  Behind `if (ok)` branches hipcc waits vmcnt(0) in front of every store (it cannot order the
  // conditional loads), i.e. one full memory round trip per store.
  const int g = lane >> 4, jr = lane & 15;
  constexpr int CW = 4 * NT;                    // consecutive columns per lane and (x, y, i)
  typedef int i32x2_t __attribute__((ext_vector_type(2)));
  auto mk = [](const void* ptr, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000); };
  const long long cbyte = coff * ((EPI & E_F32) ? 4 : 2);
  const auto rC = mk(p.C + cbyte, OOB);
  const auto rRes = mk(p.residual != nullptr ? p.residual + cbyte : nullptr, p.residual != nullptr ? OOB : 0u);
  const auto rPre = mk(p.preact != nullptr ? p.preact + cbyte : nullptr, p.preact != nullptr ? OOB : 0u);
  const auto rRelu = mk(p.relu_src != nullptr ? p.relu_src + cbyte : nullptr, p.relu_src != nullptr ? OOB : 0u);
  const auto rBias = mk(p.bias, p.bias != nullptr ? (unsigned)p.N * 4u : 0u);
  auto ldf4 = [&](const auto& rs, unsigned off, float* dst) {   // 4 floats
    const i32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    dst[0] = __int_as_float(q.x); dst[1] = __int_as_float(q.y); dst[2] = __int_as_float(q.z); dst[3] = __int_as_float(q.w);
  };
  auto ldbf = [&](const auto& rs, unsigned off, float (&dst)[CW]) {   // CW bf16 = 16 (+ 8) bytes
    const i32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    dst[0] = __uint_as_float((unsigned)q.x << 16); dst[1] = __uint_as_float((unsigned)q.x & 0xffff0000u);
    dst[2] = __uint_as_float((unsigned)q.y << 16); dst[3] = __uint_as_float((unsigned)q.y & 0xffff0000u);
    dst[4] = __uint_as_float((unsigned)q.z << 16); dst[5] = __uint_as_float((unsigned)q.z & 0xffff0000u);
    dst[6] = __uint_as_float((unsigned)q.w << 16); dst[7] = __uint_as_float((unsigned)q.w & 0xffff0000u);
    if constexpr (CW == 12) {
      const i32x2_t q2 = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 16, 0, 0);
      dst[8] = __uint_as_float((unsigned)q2.x << 16); dst[9] = __uint_as_float((unsigned)q2.x & 0xffff0000u);
      dst[10] = __uint_as_float((unsigned)q2.y << 16); dst[11] = __uint_as_float((unsigned)q2.y & 0xffff0000u);
    }
  };
  auto stbf = [&](const auto& rs, unsigned off, const float (&src)[CW]) {
    i32x4_t q;
    q.x = (int)pack_bf16x2(src[0], src[1]); q.y = (int)pack_bf16x2(src[2], src[3]);
    q.z = (int)pack_bf16x2(src[4], src[5]); q.w = (int)pack_bf16x2(src[6], src[7]);
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, off, 0, 0);
    if constexpr (CW == 12) {
      i32x2_t q2;
      q2.x = (int)pack_bf16x2(src[8], src[9]); q2.y = (int)pack_bf16x2(src[10], src[11]);
      __builtin_amdgcn_raw_buffer_store_b64(q2, rs, off + 16, 0, 0);
    }
  };
  // per-column bias: Linear layers, and the folded BatchNorm shift of an eval-mode conv forward; per-column scale: the latter only
  constexpr bool HAS_BIAS = GATHER == 0 || (EPI & E_SCALE_RELU);
  float biasv[2][HAS_BIAS ? CW : 1], scalev[2][(EPI & E_SCALE_RELU) ? CW : 1];
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int nb = n0 + y * C::HN + wc * C::SN + CW * g;
    if constexpr (HAS_BIAS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) ldf4(rBias, (unsigned)(nb + 4 * t) * 4u, &biasv[y][4 * t]);
    }
    if constexpr (EPI & E_SCALE_RELU) {
      const auto rSc = mk(p.colscale, p.colscale != nullptr ? (unsigned)p.N * 4u : 0u);
#pragma unroll
      for (int t = 0; t < NT; ++t) ldf4(rSc, (unsigned)(nb + 4 * t) * 4u, &scalev[y][4 * t]);
#pragma unroll
      for (int e = 0; e < CW; ++e) scalev[y][e] = p.colscale != nullptr ? scalev[y][e] * p.alpha : p.alpha;
    }
  }
  const float alpha = p.alpha;
  // column sums over this lane's rows.  E_CSTATS: sum a, sum a^2 of the accumulators (train-mode BatchNorm statistics).
  // E_BNB*: sum g and sum g*x_t of the stored gradient g against the raw BatchNorm inputs x_t; the centred form
  // sum g*(x_t - mean_t)*rstd_t is taken once per tile and column in the final reduction (the per-element form cost
  // 32 more registers per set, which this epilogue does not have beside 128 accumulators).
  constexpr int NBN = (EPI & E_BNB2) ? 2 : ((EPI & E_BNB1) ? 1 : 0);
  constexpr int NSUM = (EPI & E_CSTATS) ? 2 : (NBN > 0 ? 1 + NBN : 0);
  float sums[NSUM > 0 ? NSUM : 1][2][CW];
#pragma unroll
  for (int q = 0; q < (NSUM > 0 ? NSUM : 1); ++q)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int e = 0; e < CW; ++e) sums[q][y][e] = 0.f;
  const bool cls = GATHER == 2 && p.cls_h >= 0;
  const unsigned ldc = (unsigned)p.ldc;

  // Items = the wave's 2*MT row tiles (each 16 rows x 2 column halves).  The side inputs of item it+1 (residual, saved
  // pre-activation, ReLU source, BatchNorm inputs) are requested BEFORE item it is finished and stored: vmcnt counts
  // loads and stores in issue order, so a load issued behind a store could only be waited for together with that store.
  constexpr int S_RES = 0, S_PRE = (EPI & E_RES) ? 1 : 0, S_RELU = S_PRE + ((EPI & E_GELUGRAD) ? 1 : 0),
                S_BX = S_RELU + ((EPI & E_RELUMASK) ? 1 : 0), NSIDE = S_BX + NBN;
  struct Raw {
    unsigned ob[2];
    bool ok[2];
    i32x4_t q[NSIDE > 0 ? NSIDE : 1][2];
    i32x2_t q2[NSIDE > 0 ? NSIDE : 1][2];
  };
  auto unpack = [&](const Raw& r, int sidx, int y, float (&dst)[CW]) {
    const i32x4_t q = r.q[sidx][y];
    dst[0] = __uint_as_float((unsigned)q.x << 16); dst[1] = __uint_as_float((unsigned)q.x & 0xffff0000u);
    dst[2] = __uint_as_float((unsigned)q.y << 16); dst[3] = __uint_as_float((unsigned)q.y & 0xffff0000u);
    dst[4] = __uint_as_float((unsigned)q.z << 16); dst[5] = __uint_as_float((unsigned)q.z & 0xffff0000u);
    dst[6] = __uint_as_float((unsigned)q.w << 16); dst[7] = __uint_as_float((unsigned)q.w & 0xffff0000u);
    if constexpr (CW == 12) {
      const i32x2_t q2 = r.q2[sidx][y];
      dst[8] = __uint_as_float((unsigned)q2.x << 16); dst[9] = __uint_as_float((unsigned)q2.x & 0xffff0000u);
      dst[10] = __uint_as_float((unsigned)q2.y << 16); dst[11] = __uint_as_float((unsigned)q2.y & 0xffff0000u);
    }
  };
  const auto rBx0 = mk(NBN > 0 ? p.bnb_x[0] + cbyte : nullptr, NBN > 0 ? OOB : 0u);
  const auto rBx1 = mk(NBN > 1 ? p.bnb_x[1] + cbyte : nullptr, NBN > 1 ? OOB : 0u);
  auto request = [&](int it, Raw& r) {
    const int x = it / MT, i = it - x * MT;
    int m = m0 + x * C::HM + wr * C::SM + 16 * i + jr;
    const bool mok = m < p.M;
    if (cls) {  // class row -> input-pixel row of the NHWC gradient
      int b, hq, wq;
      if (p.wq_shift >= 0) {
        b = m >> p.hwq_shift;
        const int rr = m & ((1 << p.hwq_shift) - 1);
        hq = rr >> p.wq_shift;
        wq = rr & ((1 << p.wq_shift) - 1);
      } else {
        const int hw = p.Hq * p.Wq;
        b = m / hw;
        const int rr = m - b * hw;
        hq = rr / p.Wq;
        wq = rr - hq * p.Wq;
      }
      m = (b * p.Hi + hq * p.sh + p.cls_h) * p.Wi + wq * p.sw + p.cls_w;
    }
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int nb = n0 + y * C::HN + wc * C::SN + CW * g;
      r.ok[y] = mok && nb < p.N;     // N is a multiple of the 4*NT-column groups for every shape routed here (host check)
      const unsigned oe = (unsigned)m * ldc + (unsigned)nb;      // element offset (< 2^30: host check)
      const unsigned ob = r.ok[y] ? oe * ((EPI & E_F32) ? 4u : 2u) : OOB;
      r.ob[y] = ob;
      auto ldraw = [&](const auto& rs, int sidx) {
        r.q[sidx][y] = __builtin_amdgcn_raw_buffer_load_b128(rs, ob, 0, 0);
        if constexpr (CW == 12) r.q2[sidx][y] = __builtin_amdgcn_raw_buffer_load_b64(rs, ob + 16, 0, 0);
      };
      if constexpr (EPI & E_RES) ldraw(rRes, S_RES);
      if constexpr (EPI & E_GELUGRAD) ldraw(rPre, S_PRE);
      if constexpr (EPI & E_RELUMASK) ldraw(rRelu, S_RELU);
      if constexpr (NBN > 0) ldraw(rBx0, S_BX);
      if constexpr (NBN > 1) ldraw(rBx1, S_BX + 1);
    }
  };
  auto finish = [&](auto xc, auto ic, const Raw& r) {
    constexpr int x = decltype(xc)::value, i = decltype(ic)::value;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const unsigned ob = r.ob[y];
      float v[CW];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) v[4 * t + rg] = acc[x][y][i][t][rg];
      if constexpr (EPI & E_CSTATS) {
#pragma unroll
        for (int e = 0; e < CW; ++e) {   // rows >= M hold exact zeros (zero-filled operands)
          sums[0][y][e] += v[e];
          sums[1][y][e] += v[e] * v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        const float sc = (EPI & E_SCALE_RELU) ? scalev[y][(EPI & E_SCALE_RELU) ? e : 0] : alpha;
        v[e] = HAS_BIAS ? v[e] * sc + biasv[y][HAS_BIAS ? e : 0] : v[e] * sc;
      }
      if constexpr (EPI & E_F32) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          i32x4_t q = {__float_as_int(v[4 * t]), __float_as_int(v[4 * t + 1]), __float_as_int(v[4 * t + 2]), __float_as_int(v[4 * t + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(q, rC, ob + 16 * t, 0, 0);
        }
      } else {
        if constexpr (EPI & E_GELUGRAD) {
          float xp[CW];
          unpack(r, S_PRE, y, xp);
#pragma unroll
          for (int e = 0; e < CW; e += 2) {
            const f32x2_t gg = gelu_erf_grad_fast2(f32x2_t{xp[e], xp[e + 1]});
            v[e] *= gg.x;
            v[e + 1] *= gg.y;
          }
        }
        if constexpr (EPI & E_GELU) {
          // the saved pre-activation is the bf16-rounded value and GELU is taken of that rounded value, so that the
          // backward's GELU'(saved) belongs to exactly the function the forward applied
          stbf(rPre, ob, v);
#pragma unroll
          for (int e = 0; e < CW; e += 2) {
            const unsigned w = pack_bf16x2(v[e], v[e + 1]);
            const f32x2_t gg = gelu_erf_fast2(f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)});
            v[e] = gg.x;
            v[e + 1] = gg.y;
          }
        }
        if constexpr (EPI & E_RES) {
          float rr[CW];
          unpack(r, S_RES, y, rr);
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] += rr[e];
        }
        if constexpr (EPI & E_SCALE_RELU) {
          if (p.act == 3) {
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] = fmaxf(v[e], 0.f);
          }
        }
        if constexpr (EPI & E_RELUMASK) {
          float ys[CW];
          unpack(r, S_RELU, y, ys);
#pragma unroll
          for (int e = 0; e < CW; ++e) v[e] = ys[e] > 0.f ? v[e] : 0.f;
        }
        if constexpr (NBN > 0) {
          // what is summed is the gradient as STORED (rounded to bfloat16), the value the BatchNorm-backward apply pass reads
          float gq[CW];
#pragma unroll
          for (int e = 0; e < CW; e += 2) {
            const unsigned w = pack_bf16x2(v[e], v[e + 1]);
            gq[e] = r.ok[y] ? __uint_as_float(w << 16) : 0.f;
            gq[e + 1] = r.ok[y] ? __uint_as_float(w & 0xffff0000u) : 0.f;
          }
#pragma unroll
          for (int e = 0; e < CW; ++e) sums[0][y][e] += gq[e];
#pragma unroll
          for (int t = 0; t < NBN; ++t) {
            float xs[CW];
            unpack(r, S_BX + t, y, xs);
#pragma unroll
            for (int e = 0; e < CW; ++e) sums[1 + t][y][e] += gq[e] * xs[e];
          }
        }
        stbf(rC, ob, v);
      }
    }
  };
  {
    // ring of DEPTH items of side inputs in flight (the dead operand-fragment registers hold them): a load must be
    // issued ~1-2 us before its use, one item of look-ahead (~150 cycles of work) would serialise the items on it
    constexpr int NIT = 2 * MT;
    constexpr int REGS_PER = (CW / 2) * 2 * (NSIDE > 0 ? NSIDE : 1);       // VGPRs of one item's side inputs
    constexpr int BUDGET = C::BN == 256 ? 64 : 96;
    constexpr int DEPTH = NSIDE == 0 ? 1 : (BUDGET / REGS_PER >= NIT ? NIT : (BUDGET / REGS_PER < 2 ? 2 : BUDGET / REGS_PER));
    Raw rw[DEPTH];
#pragma unroll
    for (int it = 0; it < DEPTH; ++it) request(it, rw[it]);
    auto item = [&](auto itc) {
      constexpr int it = decltype(itc)::value;
      if constexpr (it < NIT) {
        __builtin_amdgcn_sched_barrier(0);
        finish(std::integral_constant<int, it / MT>{}, std::integral_constant<int, it % MT>{}, rw[it % DEPTH]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (it + DEPTH < NIT) request(it + DEPTH, rw[it % DEPTH]);
      }
    };
    item(std::integral_constant<int, 0>{});
    item(std::integral_constant<int, 1>{});
    item(std::integral_constant<int, 2>{});
    item(std::integral_constant<int, 3>{});
    item(std::integral_constant<int, 4>{});
    item(std::integral_constant<int, 5>{});
    item(std::integral_constant<int, 6>{});
    item(std::integral_constant<int, 7>{});
  }

  // ---- column sums: reduce over the 16 row lanes, then over the WARPS_M waves through LDS ----
  if constexpr (NSUM > 0) {
    float* red = reinterpret_cast<float*>(smem);       // [WARPS_M][BN][NSUM]; the operand tiles are dead
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NSUM; ++q)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          float a = sums[q][y][e];
#pragma unroll
          for (int sft = 1; sft < 16; sft <<= 1) a += __shfl_xor(a, sft, 64);
          if (jr == 0) red[(wr * C::BN + y * C::HN + wc * C::SN + CW * g + e) * NSUM + q] = a;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C::BN; c += 512) {
      const int n = n0 + c;
      if (n < p.N) {
        float tot[NSUM];
#pragma unroll
        for (int q = 0; q < NSUM; ++q) {
          tot[q] = 0.f;
#pragma unroll
          for (int w = 0; w < C::WARPS_M; ++w) tot[q] += red[(w * C::BN + c) * NSUM + q];
        }
        if constexpr (EPI & E_CSTATS) {
          float* dst = p.colstats + (long long)tile_m * 2 * p.N;
          dst[n] = tot[0];
          dst[p.N + n] = tot[1];
        } else {
#pragma unroll
          for (int t = 0; t < NBN; ++t) {
            float* dst = p.bnb_partial[t] + (long long)(p.bnb_tile0 + tile_m) * 2 * p.N;
            dst[n] = tot[0];
            dst[p.N + n] = (tot[1 + t] - p.bnb_mean[t][n] * tot[0]) * p.bnb_rstd[t][n];
          }
        }
      }
    }
  }
}

template <class C, int GATHER, int EPI>
__global__ __launch_bounds__(512) void gemm8p_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm8p_body<C, GATHER, EPI>(*kp, (int)blockIdx.x);
}

template <class C, int GATHER, int EPI>
int launch(const KParams& p, int zdim, hipStream_t st) {
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS");
  static bool attr_done = false;
  auto kern = gemm8p_kernel<C, GATHER, EPI>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", C::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, zdim);
  hipLaunchKernelGGL(kern, grid, dim3(512), C::LDS_BYTES, st, p);
  set_last_kernel("gemm8p_kernel<Cfg<%d, %d, %d>, %d, %d>", C::BN, C::WARPS_M, C::WARPS_N, GATHER, EPI);
  const int rc = check_launch("gemm8p_kernel");
  return rc ? rc : 1;
}

}  // namespace g8
