// gemm_dma.hip -- bfloat16 throughput path of htrvt_gemm: 256 x BN x 64 tiles,
// 512 threads = 8 waves (4 along M x 2 along N, each 64 x BN/2), operand tiles
// moved global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPR staging,
// no ds_write), double buffered: the DMA of k-tile t+1 is in flight while k-tile t
// is multiplied (v_mfma_f32_32x32x16_bf16), one barrier per k-tile.
//
// Why buffer loads: an out-of-range byte offset makes the hardware deliver zeros,
// so image-border taps of the implicit-GEMM convolutions (zero padding), M/N/K
// tails and strided-dgrad "holes" cost no branch -- an invalid 16-byte chunk simply
// gets the offset 0x80000000 (every operand is < 2 GiB).
//
// LDS images (DMA writes lane-linear: base + 16*lane, so any swizzle is applied to
// the per-lane SOURCE chunk and again on the read):
//   K-major operand  : [rows][128 B], 16-byte chunk c stored at c ^ ((row>>1) & 7) -> ds_read_b128, conflict-free
//                      (two 128-B rows share a 256-B bank line: the 16 rows of a ds_read_b128 lane group hit 16 slots)
//   MN-major operand : [64 k][rows*2 B], chunk c stored at (c + 4*(k&3)) mod CPR  -> ds_read_b64_tr_b16 (transpose)
#pragma once
#include <type_traits>

#include "gemm_common.h"

using namespace htrvt;

namespace {

#ifdef HTRVT_EXP_STAMP
__device__ unsigned long long htrvt_dbg[16 * 8192];   // one per translation unit (experiment builds only)
#define HTRVT_STAMP(slot)                                                                         \
  do {                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 8192) htrvt_dbg[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define HTRVT_STAMP(slot) do { } while (0)
#endif


constexpr int BK = 64;
constexpr unsigned OOB = 0x80000000u;

typedef int i32x4_t __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 B -> LDS [lds_addr, lds_addr + 1 KiB).  Inline asm on purpose: hipcc would
// otherwise order every later ds_read behind the DMA with s_waitcnt vmcnt(0) (it cannot prove the two LDS stages
// disjoint), which serialises load and MFMA.  The kernel counts these loads itself (vmcnt before the barrier).
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

// Split factors that are not multiples of 8 (layer 2: 14 ranges x 18 tiles, layer 3: 7 x 72, layer 1: 42 x 6): workgroup b runs on
// XCD b % 8, so the workgroups of XCD x are b = x, x + 8, ...; give them CONSECUTIVE (range, tile) pairs -- XCD x holds the linear
// indices [P_x, P_x + n_x) -- and every pixel range lies on one XCD or straddles two, instead of being dealt over all eight (each
// range's x / dY rows were then fetched into eight L2s: 4.8x the operand bytes at layer 2, profiles/r05_pmc.md).
__device__ __forceinline__ void xcd_range_map(int b, int total, int ntiles, int& z, int& id) {
  const int x = b & 7, s = b >> 3;
  const int q = total >> 3, r = total & 7;
  const int L = x * q + (x < r ? x : r) + s;
  z = L / ntiles;
  id = L - z * ntiles;
}

inline bool hwgrad_xcd_ranges() {      // HTRVT_NO_XCD_RANGES=1: the z-grid of rounds 3-4 for split factors that are not multiples of 8 (A/B runs)
  static const bool off = getenv("HTRVT_NO_XCD_RANGES") != nullptr && getenv("HTRVT_NO_XCD_RANGES")[0] == '1';
  return !off;
}

template <int ROWS, int NWAVES = 8, int KB = 64>   // KB = k-tile depth in elements (64, or 32 for the deep pipeline)
struct Geo {
  static constexpr int BYTES = ROWS * KB * 2;           // both layouts
  static constexpr int NP = BYTES / 1024 / NWAVES;      // 1-KiB DMA pieces per wave per k-tile
  static constexpr int CPR_MN = ROWS / 8;               // 16-byte chunks per k-row of an MN-major tile
  static constexpr int KROWB = KB * 2;                  // bytes per row of a K-major tile
  static constexpr int CPRK = KB / 8;                   // 16-byte chunks per row of a K-major tile
  static constexpr int RPP = 1024 / KROWB;              // K-major rows per DMA piece
  // XOR swizzle of the chunk index of K-major row `row`: 256 / KROWB rows share a 256-byte bank line, the 16 rows
  // of one ds_read_b128 lane group must land on 16 different 16-byte slots
  static __device__ __forceinline__ int swz(int row) {
    if constexpr (KB == 64) return (row >> 1) & 7;
    else return (row >> 2) & 3;
  }
  // rotation (in 16-byte chunks) of k-row k of an MN-major tile, chosen so that the four k-rows one
  // ds_read_b64_tr_b16 half-wave touches land on four different 64-byte bank groups:
  //   256- / 512-byte rows start on a bank-line boundary      -> 4 * (k & 3)
  //   384-byte rows start at 0 / 128 alternately (k & 1)       -> 4 * ((k >> 1) & 1)   (exhaustive search)
  static __device__ __forceinline__ int rot(int k) {
    if constexpr (ROWS == 192) return 4 * ((k >> 1) & 1);
    else if constexpr (ROWS >= 128) return 4 * (k & 3);
    else return 0;
  }
};

template <class P>
__device__ __forceinline__ void pix_decode(const P& p, int m, int& b, int& ho, int& wo) {
  if (p.howo_shift >= 0) {
    b = m >> p.howo_shift;
    const int r = m & ((1 << p.howo_shift) - 1);
    ho = r >> p.wo_shift;
    wo = r & ((1 << p.wo_shift) - 1);
  } else {
    const int hw = p.Ho * p.Wo;
    b = m / hw;
    const int r = m - b * hw;
    ho = r / p.Wo;
    wo = r - ho * p.Wo;
  }
}

// ROLE: 0 plain, 1 conv-fwd rows, 2 conv-dgrad rows (K-major), 3 conv-wgrad (MN-major, k = output pixel)
template <int ROWS, int LAYOUT, int ROLE, int NWAVES, int KB = 64>
struct DmaLoader {
  using G = Geo<ROWS, NWAVES, KB>;
  static constexpr int NP = G::NP;
  unsigned off0[NP];
  int c0[NP], c1[NP], c2[NP];
  bool ok[NP];
  i32x4_t rsrc;
  unsigned ld2;  // leading dimension in bytes
  // conv row gather: byte offset of (pixel reached through the current tap, this lane's chunk) or OOB -- refreshed
  // only when the k-loop enters a new tap, so a k-tile inside a tap costs one add per piece
  unsigned tapoff[NP];
  int cur_ti;

  template <class P>
  __device__ __forceinline__ void init(const P& p, const char* base, long long ld, int row0, int rows_total, int wave,
                                       int lane) {
    const unsigned long long ba = (unsigned long long)base;  // raw buffer, stride 0, 2 GiB of records
    rsrc = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
    ld2 = (unsigned)(ld * 2);
    cur_ti = -1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pi = wave + NWAVES * i;
      if constexpr (LAYOUT == HTRVT_KMAJOR) {
        const int rl = pi * G::RPP + lane / G::CPRK;
        const int cg = (lane % G::CPRK) ^ G::swz(rl);
        const int row = row0 + rl;
        ok[i] = row < rows_total;
        c2[i] = cg * 8;
        if constexpr (ROLE == 0) {
          off0[i] = (unsigned)row * ld2 + cg * 16;
          c0[i] = c1[i] = 0;
        } else if constexpr (ROLE == 1) {
          int b, ho, wo;
          const int hw = p.Ho * p.Wo;
          b = row / hw;
          const int r = row - b * hw;
          ho = r / p.Wo;
          wo = r - ho * p.Wo;
          off0[i] = (unsigned)b * p.Hi * p.Wi * p.Ci * 2;
          c0[i] = ho * p.sh - p.ph;
          c1[i] = wo * p.sw - p.pw;
        } else {  // input pixel of this row (all pixels, or the pixels of one stride-parity class)
          const int hw = p.Hq * p.Wq;
          const int b = row / hw, r = row - b * hw;
          int hi = r / p.Wq, wi = r - hi * p.Wq;
          if (p.cls_h >= 0) {
            hi = hi * p.sh + p.cls_h;
            wi = wi * p.sw + p.cls_w;
          }
          off0[i] = (unsigned)b * p.Ho * p.Wo * p.Co * 2;
          c0[i] = hi + p.ph;
          c1[i] = wi + p.pw;
        }
      } else {
        const int s = pi * 64 + lane;
        const int krow = s / G::CPR_MN, cl = s - krow * G::CPR_MN;
        int cg = cl - G::rot(krow);
        if (cg < 0) cg += G::CPR_MN;
        const int col = row0 + cg * 8;
        c2[i] = krow;
        if constexpr (ROLE == 0) {
          ok[i] = col < rows_total;
          off0[i] = (unsigned)col * 2;
          c0[i] = c1[i] = 0;
        } else {  // ROLE 3: col = tap*Cpad + ci
          const int tap = col / p.Cpad, ci = col - tap * p.Cpad;
          const int dy = tap / p.kw, dx = tap - dy * p.kw;
          ok[i] = (col < rows_total) && (ci < p.Ci);
          off0[i] = (unsigned)ci * 2;
          c0[i] = dy - p.ph;
          c1[i] = dx - p.pw;
        }
      }
    }
  }

  // kmap: plain K-major operand (the packed conv weights) whose k runs over a SELECTED tap list
  template <bool KMAP = false, class P = KParams>
  __device__ __forceinline__ void issue(const P& p, unsigned lds_tile, int k0, int kend, int wave) {
    int tap_dy = 0, tap_dx = 0, cbase = k0;
    if constexpr (ROLE == 1 || ROLE == 2 || KMAP) {
      const int ti = k0 / p.Cpad;
      const int tap = (int)((p.tappack >> (4 * ti)) & 15ull);   // = p.tapsel[ti], without the memory round trip
      cbase = k0 - ti * p.Cpad;
      tap_dy = tap / p.kw;
      tap_dx = tap - tap_dy * p.kw;
      unsigned xoff = 0;
      if (ROLE == 2 && tap >= p.kh * p.kw) {   // the extra tap: the 1x1 downsample gradient (A2), at the centre tap's pixel
        tap_dy = p.ph;
        tap_dx = p.pw;
        xoff = p.extra_off;
      }
      if constexpr (KMAP) cbase += tap * p.Cpad;  // column of the packed weight matrix
      if constexpr (ROLE == 1 || ROLE == 2) {
        if (ti != cur_ti) {  // wave-uniform: entering a new tap
          cur_ti = ti;
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            bool v = ok[i];
            unsigned off = off0[i];
            if constexpr (ROLE == 1) {
              const int hi = c0[i] + tap_dy, wi = c1[i] + tap_dx;
              v = v && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
              off += (unsigned)((hi * p.Wi + wi) * p.Ci + c2[i]) * 2;
            } else {
              const int th = c0[i] - tap_dy, tw = c1[i] - tap_dx;
              const int ho = th >> (p.sh - 1), wo = tw >> (p.sw - 1);
              v = v && (th >= 0) && (tw >= 0) && ((th & (p.sh - 1)) == 0) && ((tw & (p.sw - 1)) == 0) && (ho < p.Ho) &&
                  (wo < p.Wo);
              off += (unsigned)((ho * p.Wo + wo) * p.Co + c2[i]) * 2;
            }
            tapoff[i] = v ? off + xoff : OOB;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      bool v = ok[i];
      unsigned off = off0[i];
      if constexpr (LAYOUT == HTRVT_KMAJOR) {
        if constexpr (ROLE == 0) {
          v = v && (k0 + c2[i] < kend);
          off += (unsigned)(KMAP ? cbase : k0) * 2;
        } else {  // ROLE 1 / 2: cached tap offset + channel offset; an OOB base stays OOB (adds < 2^16)
          const int cvalid = ROLE == 1 ? p.Ci : p.Co;
          v = (cbase + c2[i] < cvalid);
          off = tapoff[i] + (unsigned)cbase * 2;
        }
      } else {
        const int k = k0 + c2[i];
        v = v && (k < kend);
        if constexpr (ROLE == 0) {
          off += (unsigned)k * ld2;
        } else if (p.wo_shift >= 6) {
          // Wo is a power of two >= 64 and k0 is a multiple of 64: the whole k-tile lies in ONE output row, so
          // (b, ho, wo0) are scalars and only the column varies per lane
          const int bq = k0 >> p.howo_shift, rq = k0 & ((1 << p.howo_shift) - 1);
          const int hoq = rq >> p.wo_shift, wo0 = rq & ((1 << p.wo_shift) - 1);
          const int hi = hoq * p.sh + c0[i], wi = (wo0 + c2[i]) * p.sw + c1[i];
          v = v && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          off += (unsigned)(((bq * p.Hi + hi) * p.Wi + wi) * p.Ci) * 2;
        } else {
          int b, ho, wo;
          pix_decode(p, k, b, ho, wo);
          const int hi = ho * p.sh + c0[i], wi = wo * p.sw + c1[i];
          v = v && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          off += (unsigned)(((b * p.Hi + hi) * p.Wi + wi) * p.Ci) * 2;
        }
      }
      const unsigned voff = (v && k0 < kend) ? off : OOB;   // k0 >= kend: a zero-fill piece past the last k-tile
      dma16(rsrc, __builtin_amdgcn_readfirstlane(lds_tile + (wave + NWAVES * i) * 1024), voff);
    }
  }
};

template <int ROWS, int LAYOUT, int KB = 64>
__device__ __forceinline__ bf16x8_t frag_read(const char* lds, int rb, int s, int lane) {
  using G = Geo<ROWS, 8, KB>;
  if constexpr (LAYOUT == HTRVT_KMAJOR) {
    const int row = rb * 32 + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * G::KROWB + ((chunk ^ G::swz(row)) << 4));
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
    int chunk = rb * 4 + 2 * (g & 1) + (pp >> 1) + G::rot(q);
    if (chunk >= G::CPR_MN) chunk -= G::CPR_MN;
    const int krow = 16 * s + 8 * h + q;
    const char* a0 = lds + krow * (ROWS * 2) + chunk * 16 + (pp & 1) * 8;
    typedef __attribute__((address_space(3))) s16x4_t* lptr;
    const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
    const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * (ROWS * 2)));
    const s16x8_t r = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 epilogue staged through LDS: accumulators -> (alpha, bias) -> bf16, written column-major into the dead
// operand buffers; read back transposed (ds_read_b64_tr_b16) so that every lane owns 8 consecutive columns of
// one row: GELU / GELU' / residual / pre-activation traffic and the C store are all 16-byte, row-contiguous.
// (A per-lane 2-byte store epilogue is store-issue bound: 96 store instructions per wave for a 64x96 block.)
// ---------------------------------------------------------------------------------------------
template <int BM>
struct Stg {
  static constexpr int CST = BM * 2 + 8;  // bytes per staged column (BM rows + pad: conflict-free ds_write_b64)
};

__device__ __forceinline__ float bf16lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// DGRAD: compile the backward-of-ReLU / BatchNorm-backward-sum path (conv-dgrad kernels only: it costs registers)
// CSTATS: compile the per-column sum / sum-of-squares path (conv-forward kernels only: BatchNorm batch statistics)
// MF: the MFMA shape the accumulators come from.  32: acc[i][j][r] = C(32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), 32 j + (lane & 31)).
// 16 (v_mfma_f32_16x16x32_bf16, four 16 x 16 tiles per 32 x 32 block): acc[i][j][4 (2 a + b) + r] = C(32 i + 16 a + 4 (lane >> 4) + r,
// 32 j + 16 b + (lane & 15))
template <int TN, int BN, int BM, int NW_TOTAL, bool DGRAD, bool CSTATS, class P, int MF = 32>   // CSTATS (conv forward) also enables colscale / ReLU-last
__device__ __forceinline__ void epilogue_staged(f32x16_t (&acc)[2][TN], const P& p, long long coff, int m0, int n0,
                                                int wm, int wn, int tile_m, int lane, int wave, char* smem, bool active,
                                                const int rmul = 1) {
  // rmul: C row of tile row r is m0 + rmul * r (the merged strided-dgrad kernel, gemm_halo_impl.h: an M tile is 256 pixels
  // of ONE parity class of an image row, i.e. every sw-th pixel); the side inputs follow the same rows
  constexpr int CST = Stg<BM>::CST;
  constexpr int NWAVES = NW_TOTAL, NTH = NW_TOTAL * 64, NWM = BM / 64;
  // tiles wider than 192 columns are staged in two passes (the column-major staging image must fit the LDS)
  constexpr int PASSES = BN > 192 ? 2 : 1;
  constexpr int TNP = TN / PASSES, BNS = BN / PASSES;
  const int h = lane >> 5, cl = lane & 31;
  float cs1[TN], cs2[TN];        // MF = 32: sums of column 32 j + cl over this lane's rows
  float ds1[TN][2], ds2[TN][2];  // MF = 16: of columns 32 j + 16 b + (lane & 15)
#pragma unroll
  for (int j = 0; j < TN; ++j) cs1[j] = cs2[j] = ds1[j][0] = ds1[j][1] = ds2[j][0] = ds2[j][1] = 0.f;
  constexpr int GROUPS = BNS / 32;            // groups of 4 chunks (32 columns) per row
  constexpr int ITEMS = (BM / 16) * GROUPS;   // wave-level items: 16 rows x 32 columns
  const int lr = lane & 15, lg = lane >> 4;
  const int q = lr >> 2, pp = lr & 3;
  typedef __attribute__((address_space(3))) s16x4_t* lptr;
  // When the wave count is a multiple of the column groups every wave keeps ONE column group and walks the row
  // blocks: a lane then owns 8 fixed columns, which lets it accumulate per-column BatchNorm-backward sums.
  constexpr bool FIXED_COLS = (NWAVES % GROUPS) == 0;
  constexpr int RB_STEP = FIXED_COLS ? NWAVES / GROUPS : 1;
  const bool bnb = DGRAD && FIXED_COLS && PASSES == 1 && p.bnb_partial[0] != nullptr;
  const bool bnb2 = bnb && p.bnb_partial[1] != nullptr;
  // per-lane column sums of this workgroup's rows: sum g and the RAW sum g * x; the centring (x - mean) * rstd is applied
  // once per column when the tile's partial row is written (two VALU operations and 16 registers per set less in the walk)
  float bs1[2][8], bs2[2][8];
  // ReLU mask recomputed from bnb_x[0]: the forward's BatchNorm scale / shift of this lane's 8 columns
  const bool relux = bnb && !bnb2 && p.relu_sc != nullptr;
  float rsc[8], rsf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) rsc[e] = rsf[e] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bs1[t][e] = bs2[t][e] = 0.f;
  if (relux) {
    const int nc = n0 + (wave % GROUPS) * 32 + lg * 8;
    if (nc < p.N) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        rsc[e] = p.relu_sc[nc + e];
        rsf[e] = p.relu_sf[nc + e];
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
  if (pass > 0) __syncthreads();   // the previous pass has drained the staging image
  // ---- phase 1: registers -> LDS (column-major bf16); loader waves hold no accumulators ----
  if (active) {
  float biasv[TNP], scalev[TNP];
#pragma unroll
  for (int jp = 0; jp < TNP; ++jp) {
    biasv[jp] = 0.f;
    scalev[jp] = p.alpha;
  }
  if (p.bias != nullptr) {   // all loads of the pass in flight together (clamped index: no per-column branch)
#pragma unroll
    for (int jp = 0; jp < TNP; ++jp) {
      const int n = n0 + (wn * TN + pass * TNP + jp) * 32 + cl;
      const float b = p.bias[n < p.N ? n : 0];
      biasv[jp] = n < p.N ? b : 0.f;
    }
  }
  if (CSTATS && p.colscale != nullptr) {   // eval-mode BatchNorm folded into the convolution: per-column scale (and bias = shift)
#pragma unroll
    for (int jp = 0; jp < TNP; ++jp) {
      const int n = n0 + (wn * TN + pass * TNP + jp) * 32 + cl;
      scalev[jp] = p.alpha * p.colscale[n < p.N ? n : 0];
    }
  }
  if constexpr (MF == 16) {
    // this lane's columns are 32 j + 16 b + (lane & 15): its own bias / scale values (the loads above were for column 32 j + cl)
    float bias16[TNP][2], scale16[TNP][2];
#pragma unroll
    for (int jp = 0; jp < TNP; ++jp)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = n0 + (wn * TN + pass * TNP + jp) * 32 + 16 * b + (lane & 15);
        const int nc = n < p.N ? n : 0;
        bias16[jp][b] = (p.bias != nullptr && n < p.N) ? p.bias[nc] : 0.f;
        scale16[jp][b] = (CSTATS && p.colscale != nullptr) ? p.alpha * p.colscale[nc] : p.alpha;
      }
#pragma unroll
    for (int jp = 0; jp < TNP; ++jp) {
      const int j = pass * TNP + jp;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int ccol = (wn * TNP + jp) * 32 + 16 * b + (lane & 15);
            const int crow = (wm * 2 + i) * 32 + 16 * a + 4 * (lane >> 4);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = acc[i][j][4 * (2 * a + b) + r];
              if constexpr (CSTATS) {
                ds1[j][b] += x;
                ds2[j][b] += x * x;
              }
              v[r] = x * scale16[jp][b] + bias16[jp][b];
            }
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(smem + ccol * CST + crow * 2) = o;
          }
    }
  } else
#pragma unroll
  for (int jp = 0; jp < TNP; ++jp) {
    const int j = pass * TNP + jp;
    const int ccol = (wn * TNP + jp) * 32 + cl;          // column in the staging image
    const float bias = biasv[jp], al = scalev[jp];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int crow = (wm * 2 + i) * 32 + 8 * g + 4 * h;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = acc[i][j][4 * g + r];  // rows >= M and cols >= N hold exact zeros (zero-filled operands)
          if constexpr (CSTATS) {
            cs1[j] += a;
            cs2[j] += a * a;
          }
          v[r] = a * al + bias;
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(smem + ccol * CST + crow * 2) = o;
      }
    }
  }
  }
  HTRVT_STAMP(4);
  __syncthreads();
  HTRVT_STAMP(5);
  // ---- phase 2: LDS -> (act / residual) -> global, 16 B per lane, 4 lanes = 64 contiguous bytes of one row ----
  // Items are processed U at a time: all global loads of the U items (residual, ReLU source, BN inputs, saved
  // pre-activation) are issued first from clamped, always-valid offsets, then the transposed LDS reads, then the
  // arithmetic and the stores -- otherwise every item pays a full memory round trip in sequence.
  const bool rt_res = p.residual != nullptr, rt_relu = DGRAD && p.relu_src != nullptr, rt_pre_in = !DGRAD && p.act == 2;
  const bool rt_pre_out = !rt_pre_in && p.preact != nullptr, rt_gelu = p.act == 1, rt_rlast = CSTATS && p.act == 3;
  const int rt_nb = bnb2 ? 2 : (bnb ? 1 : 0);
  const bool cls = p.cls_h >= 0;
  // Item walk: `wave` is wave-uniform, so the item's row block / column group and everything derived from them
  // live in SGPRs; a lane adds its constant part.  Offsets are 32-bit element counts (operands < 2 GiB).
  const unsigned ldc = (unsigned)p.ldc;
  bf16_t* const Cb = reinterpret_cast<bf16_t*>(p.C) + coff;
  const bf16_t* const resb = reinterpret_cast<const bf16_t*>(p.residual) + coff;
  const bf16_t* const relub = reinterpret_cast<const bf16_t*>(p.relu_src) + coff;
  const unsigned char* const relubits = reinterpret_cast<const unsigned char*>(p.relu_src) + (coff >> 3);   // the 1-bit form (p.relu_bits)
  bf16_t* const preb = reinterpret_cast<bf16_t*>(p.preact) + coff;
  const bf16_t* const bx0 = reinterpret_cast<const bf16_t*>(p.bnb_x[0]) + coff;
  const bf16_t* const bx1 = reinterpret_cast<const bf16_t*>(p.bnb_x[1]) + coff;
  const unsigned lane_lds = (lg * 8 + q) * CST + (4 * pp) * 2;

  // The walk itself, with each epilogue feature either compiled in (1), compiled out (0) or decided at run time (2).
  // The hot combinations are instantiated with constants: their side loads then sit in straight-line code and are all
  // in flight together (behind run-time branches hipcc puts an `s_waitcnt vmcnt(0)` in front of every one of them,
  // i.e. one full memory round trip per load and item).
  auto walk = [&](auto f_res, auto f_relu, auto f_pre_in, auto f_pre_out, auto f_gelu, auto f_nb, auto f_rlast) {
    constexpr int FRES = decltype(f_res)::value, FRELU = decltype(f_relu)::value, FPIN = decltype(f_pre_in)::value;
    constexpr int FPOUT = decltype(f_pre_out)::value, FGELU = decltype(f_gelu)::value, FNB = decltype(f_nb)::value;
    constexpr int FRLAST = decltype(f_rlast)::value;
    // items in flight per wave: 4, or 2 where two BatchNorm sum sets (64 accumulators), the GELU' path or the
    // run-time-flag fallback leave no registers for more under the 12-wave (168 VGPR) budget
    constexpr bool RUNTIME_FLAGS = FRES == 2 || FRELU == 2 || FPIN == 2 || FPOUT == 2 || FGELU == 2 || FRLAST == 2 || FNB == 3;
    constexpr int U = ((DGRAD && FNB >= 3) || (NW_TOTAL > 8 && (RUNTIME_FLAGS || FPIN == 1))) ? 2 : 4;
    const bool has_res = FRES == 1 || (FRES == 2 && rt_res);
    const bool has_relu = FRELU == 1 || (FRELU == 2 && rt_relu);
    constexpr bool RELU_X = FRELU == 3;       // mask from bnb_x[0] through (rsc, rsf); needs FNB >= 1
    constexpr bool RELU_BITS = FRELU == 4;    // one mask byte per item and lane instead of 16 bytes of the activation
    const bool has_pre_in = FPIN == 1 || (FPIN == 2 && rt_pre_in);
    const bool has_pre_out = FPOUT == 1 || (FPOUT == 2 && rt_pre_out);
    const bool has_gelu = FGELU == 1 || (FGELU == 2 && rt_gelu);
    const int nb = FNB == 3 ? rt_nb : FNB;
    const bool relu_last = FRLAST == 1 || (FRLAST == 2 && rt_rlast);
    const bool ew = has_res || has_relu || RELU_X || RELU_BITS || has_pre_in || has_gelu || nb > 0 || relu_last;   // any arithmetic on the staged values
    // ReLU-from-BatchNorm-input variant (one side input per item): the x vectors of BOTH rounds are requested before the
    // first round is processed -- one exposed memory round trip per tile instead of two (the accumulators are dead here, the
    // 32 registers are free).  Items of round 0 / 1 sit in two named sets, selected per round (no dynamic register index).
    constexpr bool PRE = FRELU == 3 && FNB == 1 && ITEMS == NWAVES * 2 * U;
    uint4 pbx0[PRE ? U : 1], pbx1[PRE ? U : 1];
    if constexpr (PRE) {
#pragma unroll
      for (int k = 0; k < 2 * U; ++k) {
        const int id = wave + k * NWAVES;
        const int rbk = id / GROUPS, cg = id - rbk * GROUPS;
        const int half = (cg * 32) / (TNP * 32);
        const int nbc = n0 + (half * TN + pass * TNP) * 32 + (cg * 32 - half * TNP * 32) + lg * 8;
        const int m = m0 + rmul * (rbk * 16 + lr);
        const unsigned off = (m < p.M && nbc < p.N) ? (unsigned)m * ldc + (unsigned)nbc : 0u;
        const uint4 v = *reinterpret_cast<const uint4*>(bx0 + off);
        if (k < U) pbx0[k % U] = v;
        else pbx1[k % U] = v;
      }
    }
    for (int id0 = wave; id0 < ITEMS; id0 += NWAVES * U) {
      unsigned o[U];
      bool ok[U];
      uint4 rres[U], rrelu[U], rpre[U], rbx[2][U], raw[U];
      unsigned rbits[U];
      const bool first_round = id0 == wave;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = id0 + u * NWAVES;
        const bool live = id < ITEMS;
        const int idc = live ? id : wave;           // a tail item re-reads the wave's first item and stores nothing
        const int rbk = idc / GROUPS, cg = idc - rbk * GROUPS;   // (FIXED_COLS: cg == wave % GROUPS for every item)
        const int half = (cg * 32) / (TNP * 32);    // which wave column (wn) staged this column group
        const int nbc = n0 + (half * TN + pass * TNP) * 32 + (cg * 32 - half * TNP * 32) + lg * 8;
        int m = m0 + rmul * (rbk * 16 + lr);
        ok[u] = live && m < p.M && nbc < p.N;
        if (cls) {  // class row -> input-pixel row of the NHWC gradient
          int b, hq, wq;
          if (p.wq_shift >= 0) {
            b = m >> p.hwq_shift;
            const int r = m & ((1 << p.hwq_shift) - 1);
            hq = r >> p.wq_shift;
            wq = r & ((1 << p.wq_shift) - 1);
          } else {
            const int hw = p.Hq * p.Wq;
            b = m / hw;
            const int r = m - b * hw;
            hq = r / p.Wq;
            wq = r - hq * p.Wq;
          }
          m = (b * p.Hi + hq * p.sh + p.cls_h) * p.Wi + wq * p.sw + p.cls_w;
        }
        o[u] = ok[u] ? (unsigned)m * ldc + (unsigned)nbc : 0u;   // element 0 is a valid address of every operand
        if (has_res) rres[u] = *reinterpret_cast<const uint4*>(resb + o[u]);
        if (has_relu) rrelu[u] = *reinterpret_cast<const uint4*>(relub + o[u]);
        if constexpr (RELU_BITS) rbits[u] = relubits[o[u] >> 3];
        if (has_pre_in) rpre[u] = *reinterpret_cast<const uint4*>(preb + o[u]);
        if constexpr (PRE) {
          const uint4 a_ = pbx0[u], b_ = pbx1[u];
          rbx[0][u] = first_round ? a_ : b_;
        } else if (nb > 0) {
          rbx[0][u] = *reinterpret_cast<const uint4*>(bx0 + o[u]);
        }
        if (nb > 1) rbx[1][u] = *reinterpret_cast<const uint4*>(bx1 + o[u]);
        const char* a0 = smem + (cg * 32) * CST + rbk * 32 + lane_lds;
        const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
        const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * CST));
        raw[u] = __builtin_bit_cast(uint4, s16x8_t{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w});
      }
      if (!ew) {   // workgroup-uniform: the staged bf16 values are the result
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (ok[u]) {
            if (has_pre_out) *reinterpret_cast<uint4*>(preb + o[u]) = raw[u];
            *reinterpret_cast<uint4*>(Cb + o[u]) = raw[u];
          }
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[8] = {bf16lo(raw[u].x), bf16hi(raw[u].x), bf16lo(raw[u].y), bf16hi(raw[u].y),
                      bf16lo(raw[u].z), bf16hi(raw[u].z), bf16lo(raw[u].w), bf16hi(raw[u].w)};
        if (has_pre_in) {
          const uint4 pr = rpre[u];
          const float x[8] = {bf16lo(pr.x), bf16hi(pr.x), bf16lo(pr.y), bf16hi(pr.y),
                              bf16lo(pr.z), bf16hi(pr.z), bf16lo(pr.w), bf16hi(pr.w)};
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2_t gg = gelu_erf_grad_fast2(f32x2_t{x[e], x[e + 1]});
            v[e] *= gg.x;
            v[e + 1] *= gg.y;
          }
        } else if (has_pre_out) {
          if (ok[u]) *reinterpret_cast<uint4*>(preb + o[u]) = raw[u];
        }
        if (has_gelu) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2_t gg = gelu_erf_fast2(f32x2_t{v[e], v[e + 1]});
            v[e] = gg.x;
            v[e + 1] = gg.y;
          }
        }
        if (has_res) {
          const uint4 rr = rres[u];
          v[0] += bf16lo(rr.x); v[1] += bf16hi(rr.x); v[2] += bf16lo(rr.y); v[3] += bf16hi(rr.y);
          v[4] += bf16lo(rr.z); v[5] += bf16hi(rr.z); v[6] += bf16lo(rr.w); v[7] += bf16hi(rr.w);
        }
        if (relu_last) {  // forward ReLU after the residual (eval-mode conv + BatchNorm + ReLU in one launch)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (has_relu) {  // backward of ReLU: the producer's output decides which gradients pass
          const uint4 rs = rrelu[u];
          const float y[8] = {bf16lo(rs.x), bf16hi(rs.x), bf16lo(rs.y), bf16hi(rs.y),
                              bf16lo(rs.z), bf16hi(rs.z), bf16lo(rs.w), bf16hi(rs.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = y[e] > 0.f ? v[e] : 0.f;
        }
        if constexpr (RELU_BITS) {  // the same decision, one bit per element (written by htrvt_bn_apply_mask in the forward pass)
          const unsigned rb = rbits[u];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((rb >> e) & 1u) ? v[e] : 0.f;
        }
        if constexpr (RELU_X) {  // the same ReLU, its input rebuilt from the BatchNorm input that is loaded for the sums anyway
          const uint4 xr = rbx[0][u];
          const float x[8] = {bf16lo(xr.x), bf16hi(xr.x), bf16lo(xr.y), bf16hi(xr.y),
                              bf16lo(xr.z), bf16hi(xr.z), bf16lo(xr.w), bf16hi(xr.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(x[e], rsc[e], rsf[e]) > 0.f ? v[e] : 0.f;
        }
        if (nb > 0) {  // train-mode BatchNorm backward sums of the layer this gradient feeds: sum g, sum g * xhat
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t >= nb) break;
            const uint4 xr = rbx[t][u];
            const float x[8] = {bf16lo(xr.x), bf16hi(xr.x), bf16lo(xr.y), bf16hi(xr.y),
                                bf16lo(xr.z), bf16hi(xr.z), bf16lo(xr.w), bf16hi(xr.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gv = ok[u] ? v[e] : 0.f;
              bs1[t][e] += gv;
              bs2[t][e] = fmaf(gv, x[e], bs2[t][e]);
            }
          }
        }
        uint4 out;
        out.x = pack_bf16x2(v[0], v[1]);
        out.y = pack_bf16x2(v[2], v[3]);
        out.z = pack_bf16x2(v[4], v[5]);
        out.w = pack_bf16x2(v[6], v[7]);
        if (ok[u]) *reinterpret_cast<uint4*>(Cb + o[u]) = out;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  if constexpr (DGRAD) {   // conv dgrad: [residual] [ReLU mask + 1 or 2 BatchNorm-backward sum sets]
    if (relux) walk(I0{}, I3{}, I0{}, I0{}, I0{}, I1{}, I0{});      // host guarantees: one bnb set, no residual, no relu_src
    else if (rt_relu && p.relu_bits) {   // host guarantees: relu_src + (1 set | residual + 1 set | residual + 2 sets), nothing else
      if (!rt_res) walk(I0{}, I4{}, I0{}, I0{}, I0{}, I1{}, I0{});
      else if (rt_nb == 1) walk(I1{}, I4{}, I0{}, I0{}, I0{}, I1{}, I0{});
      else walk(I1{}, I4{}, I0{}, I0{}, I0{}, I2{}, I0{});
    }
    else if (!rt_res && !rt_relu && rt_nb == 0 && !rt_rlast) walk(I0{}, I0{}, I0{}, I0{}, I0{}, I0{}, I0{});
    else if (rt_res && !rt_relu && rt_nb == 0 && !rt_rlast) walk(I1{}, I0{}, I0{}, I0{}, I0{}, I0{}, I0{});
    else if (!rt_res && rt_relu && rt_nb == 1 && !rt_rlast) walk(I0{}, I1{}, I0{}, I0{}, I0{}, I1{}, I0{});
    else if (rt_res && rt_relu && rt_nb == 1 && !rt_rlast) walk(I1{}, I1{}, I0{}, I0{}, I0{}, I1{}, I0{});
    else if (rt_res && rt_relu && rt_nb == 2 && !rt_rlast) walk(I1{}, I1{}, I0{}, I0{}, I0{}, I2{}, I0{});
    else walk(I2{}, I2{}, I0{}, I0{}, I0{}, I3{}, I2{});
  } else {                 // linear / conv forward / attention: [residual] [ReLU] | GELU [+ saved pre-activation] | * GELU'
    if (!rt_res && !rt_pre_in && !rt_pre_out && !rt_gelu && !rt_rlast) walk(I0{}, I0{}, I0{}, I0{}, I0{}, I0{}, I0{});
    else if (rt_res && !rt_pre_in && !rt_pre_out && !rt_gelu && !rt_rlast) walk(I1{}, I0{}, I0{}, I0{}, I0{}, I0{}, I0{});
    else if (!rt_res && rt_gelu && rt_pre_out) walk(I0{}, I0{}, I0{}, I1{}, I1{}, I0{}, I0{});
    else if (!rt_res && rt_pre_in) walk(I0{}, I0{}, I1{}, I0{}, I0{}, I0{}, I0{});
    else if (!rt_res && rt_rlast && !rt_pre_out) walk(I0{}, I0{}, I0{}, I0{}, I0{}, I0{}, I1{});
    else if (rt_res && rt_rlast && !rt_pre_out) walk(I1{}, I0{}, I0{}, I0{}, I0{}, I0{}, I1{});
    else walk(I2{}, I0{}, I2{}, I2{}, I2{}, I0{}, I2{});
  }
  }  // pass
  if (CSTATS && p.colstats != nullptr) {
    float* red = reinterpret_cast<float*>(smem + BNS * CST);  // [NWM][BN][2], behind the staged tile
    if constexpr (MF == 16) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float s1 = ds1[j][b], s2 = ds2[j][b];
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (lane < 16 && active) {
            const int c = (wn * TN + j) * 32 + 16 * b + lane;
            red[(wm * BN + c) * 2 + 0] = s1;
            red[(wm * BN + c) * 2 + 1] = s2;
          }
        }
    } else
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s1 = cs1[j] + __shfl_xor(cs1[j], 32, 64);
      const float s2 = cs2[j] + __shfl_xor(cs2[j], 32, 64);
      if (h == 0 && active) {
        const int c = (wn * TN + j) * 32 + cl;
        red[(wm * BN + c) * 2 + 0] = s1;
        red[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < BN; c += NTH) {
      const int n = n0 + c;
      if (n < p.N) {
        float* dst = p.colstats + (long long)tile_m * 2 * p.N;
        float a = 0.f, q2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) {
          a += red[(w * BN + c) * 2];
          q2 += red[(w * BN + c) * 2 + 1];
        }
        dst[n] = a;
        dst[p.N + n] = q2;
      }
    }
  }
  if constexpr (FIXED_COLS) {
    if (bnb) {  // workgroup-uniform
      float* red2 = reinterpret_cast<float*>(smem + BNS * CST + NWM * BN * 8);  // [2 sets][RB_STEP][BN][2]
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int sft = 1; sft < 16; sft <<= 1) {
            bs1[t][e] += __shfl_xor(bs1[t][e], sft, 64);
            bs2[t][e] += __shfl_xor(bs2[t][e], sft, 64);
          }
        }
      if (lr == 0) {
        const int c0 = (wave % GROUPS) * 32 + lg * 8, slab = wave / GROUPS;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            red2[((t * RB_STEP + slab) * BN + c0 + e) * 2 + 0] = bs1[t][e];
            red2[((t * RB_STEP + slab) * BN + c0 + e) * 2 + 1] = bs2[t][e];
          }
      }
      __syncthreads();
      for (int c = threadIdx.x; c < BN; c += NTH) {
        const int n = n0 + c;
        if (n < p.N) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t == 1 && !bnb2) break;
            float a = 0.f, q2 = 0.f;
#pragma unroll
            for (int sl = 0; sl < RB_STEP; ++sl) {
              a += red2[((t * RB_STEP + sl) * BN + c) * 2];
              q2 += red2[((t * RB_STEP + sl) * BN + c) * 2 + 1];
            }
            float* dst = p.bnb_partial[t] + (long long)(p.bnb_tile0 + tile_m) * 2 * p.N;
            dst[n] = a;
            dst[p.N + n] = (q2 - p.bnb_mean[t][n] * a) * p.bnb_rstd[t][n];     // sum g * (x - mean) * rstd
          }
        }
      }
    }
  }
}

// SPEC = 0: every wave issues its share of the DMA, then multiplies (NTH = 2*BM).
// SPEC = 1: wave specialisation -- 4 extra loader waves own ALL address arithmetic + DMA issue of the next k-tile
//           while the BM/32 consumer waves run nothing but ds_read + MFMA; one workgroup barrier per k-tile.
// NSTAGE = 3 (tiles of at most 128 columns: 3 x 48 KB of LDS): the DMA of k-tile t+2 is issued before k-tile t+1 has
// landed, so two k-tiles are in flight while one is multiplied.  With two stages the next DMA cannot start before
// the barrier that follows the previous one's arrival, and the loop is bound by one DMA round trip per k-tile
// (measured with the MFMAs removed: 2 400-3 300 cycles per k-tile against 1 900 for the MFMA side alone).
template <int BM, int BN, int AL, int BL, int GATHER, int SPEC, int NSTAGE, class P>
__device__ __forceinline__ void gemm_dma_body(const P& p, const int block_x) {
  using T = bf16_t;
  constexpr int NWC = BM / 32;                 // consumer (MFMA) waves
  constexpr int NWL = SPEC ? 4 : NWC;          // waves that issue DMA
  constexpr int NW_TOTAL = NWC + (SPEC ? 4 : 0);
  constexpr int NTH = NW_TOTAL * 64;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int A_BYTES = Geo<BM>::BYTES, B_BYTES = Geo<BN>::BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  HTRVT_STAMP(0);
  const int ntiles = p.tiles_m * p.tiles_n;
  int id = block_x;
  int z = blockIdx.z;
  if (p.split_k > 1 && (p.split_k & 7) == 0) {
    // Split-K by a multiple of 8 (weight gradients): the output tiles that share one K range read the SAME operand rows (for a conv, the
    // same pixels through different taps), so they must sit on ONE XCD at the same time to share its L2.  Blocks are
    // dealt round-robin over the 8 XCDs: linear block L -> chunk c of 8*ntiles blocks, XCD slot x = L % 8 owns K range
    // 8c + x and runs its ntiles tiles back to back.  (Measured before this mapping: the layer-1 conv wgrad fetched
    // 5.7 GB per launch against 0.8 GB of operands -- every tile streamed the pixels from HBM on a different XCD.)
    const int L = block_x;
    const int chunk = L / (8 * ntiles), r = L - chunk * 8 * ntiles;
    z = chunk * 8 + (r & 7);
    id = r >> 3;
  } else if (p.split_k > 1 && gridDim.z == 1) {     // any other split factor: consecutive (range, tile) pairs per XCD
    xcd_range_map(block_x, (int)gridDim.x, ntiles, z, id);
  } else if (p.split_k == 1 && (ntiles & 7) == 0) {
    id = (id & 7) * (ntiles >> 3) + (id >> 3);
  }
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const char* Ab = p.A;
  const char* Bb = p.B;
  int kbeg = 0, kend = p.K;
  long long coff = 0;
  if (p.split_k > 1) {
    kbeg = z * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
    coff = (long long)z * p.slab_stride;
  } else {
    const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
    Ab += (zo * p.sA_o + zi * p.sA_i) * 2;
    Bb += (zo * p.sB_o + zi * p.sB_i) * 2;
    coff = zo * p.sC_o + zi * p.sC_i;
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < NWC;
  const bool loader = SPEC ? !consumer : true;
  const int lw = SPEC ? (wave - NWC) & 3 : wave;   // index among the DMA-issuing waves
  const int wm = (wave >> 1) % (BM / 64), wn = wave & 1;
  if (SPEC && !consumer) __builtin_amdgcn_s_setprio(3);   // the loaders' issue latency is the critical path of every k-tile

  DmaLoader<BM, AL, GATHER, NWL> la;
  DmaLoader<BN, BL, 0, NWL> lb;
  la.init(p, Ab, p.lda, m0, p.M, lw, lane);  // unconditional: the descriptors must stay provably wave-uniform (SGPRs)
  lb.init(p, Bb, p.ldb, n0, p.N, lw, lane);

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (kend - kbeg + BK - 1) / BK;
  const unsigned lds0 = lds_addr_of(smem);
  constexpr bool KMAP = (GATHER == 1 || GATHER == 2);
  // every DMA-issuing wave issues exactly NP_A + NP_B pieces per k-tile, so "all but the youngest pieces have landed"
  // is a counted s_waitcnt vmcnt for it (consumer-only waves have nothing outstanding)
  constexpr int NP_A = DmaLoader<BM, AL, GATHER, NWL>::NP, NP_B = DmaLoader<BN, BL, 0, NWL>::NP;
  constexpr int PER_TILE = NP_A + NP_B;
  static_assert(NSTAGE == 2 || NSTAGE == 3 || NSTAGE == 5, "2, 3, or 3 (A) + 2 (B) LDS stages");
  static_assert(PER_TILE <= 63, "vmcnt immediate");
  // LDS: [AST buffers of A][BST buffers of B]
  constexpr int AST = NSTAGE == 5 ? 3 : NSTAGE, BST = NSTAGE == 5 ? 2 : NSTAGE;
  constexpr unsigned B_BASE = AST * A_BYTES;
  // NSTAGE == 5, the 256x192 tile's budget (3 x 32 KB + 2 x 24 KB = 144 KB): A runs two k-tiles ahead, B one.  Per
  // iteration the loaders issue B(t+1) first, then A(t+2); before the barrier everything but the youngest NP_A pieces
  // (= A(t+2)) must have landed, so only B's 24 KB are latency-critical and A(t+2) flies across the barrier.
  // Measured (dispatch<256, 192, SPEC, 5>, not instantiated in the shipped library): within +-2 % of two stages on every
  // shape of the step -- the round trip of the ONE tile that is not two ahead still bounds the loop; only the full
  // third stage (NSTAGE == 3, tiles of <= 128 columns) pays.
  if (loader) {
    if (nkt > 0) {
      la.issue(p, lds0, kbeg, kend, lw);
      lb.template issue<KMAP>(p, lds0 + B_BASE, kbeg, kend, lw);
    }
    if (AST == 3 && nkt > 1) {
      la.issue(p, lds0 + A_BYTES, kbeg + BK, kend, lw);
      if (BST == 3) lb.template issue<KMAP>(p, lds0 + B_BASE + B_BYTES, kbeg + BK, kend, lw);
    }
  }
  HTRVT_STAMP(1);
  if (NSTAGE == 3 && nkt > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
  else if (NSTAGE == 5 && nkt > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP_A) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  HTRVT_STAMP(2);

  int a_cur = 0, a_nxt = AST - 1;   // A buffer being multiplied / A buffer the DMA issued in this iteration fills
  int b_cur = 0, b_nxt = BST - 1;
  // STAGGER (every wave loads AND multiplies, two LDS stages: the weight-gradient products): with all eight waves in
  // lockstep behind the k-tile's barrier they all issue their DMA pieces first -- ~700 cycles in which no wave has an
  // MFMA in the pipe -- and then all multiply.  Ablation on the layer-1 conv weight gradient: MFMA side alone 0.52 ms, DMA
  // alone 0.57 ms, together 0.80 ms.  Waves 4-7 (the SIMD partners of waves 0-3) therefore run HALF A K-TILE BEHIND: they
  // read the fragments of a k-tile's last two k-steps into registers before the barrier and multiply them after it,
  // while waves 0-3 issue their DMA; then they issue theirs while waves 0-3 multiply.  (MI355X_MICROARCH.md, Two waves per
  // SIMD, item 9: split by wave number >= 4.)
#ifdef HTRVT_EXP_NOSTAGGER
  constexpr bool STAGGER = false;
#else
  constexpr bool STAGGER = SPEC == 0 && NSTAGE == 2 && AL == HTRVT_MNMAJOR;
#endif
  const bool late = STAGGER && wave >= NWC / 2;
  auto advance = [&]() {
    a_cur = (a_cur + 1 == AST) ? 0 : a_cur + 1;
    a_nxt = (a_nxt + 1 == AST) ? 0 : a_nxt + 1;
    b_cur = (b_cur + 1 == BST) ? 0 : b_cur + 1;
    b_nxt = (b_nxt + 1 == BST) ? 0 : b_nxt + 1;
  };
  if (late) {
    // ---- waves 4-7 of a staggered kernel: one barrier per iteration, like the loop below ----
    bf16x8_t ha0[TM], ha1[TM], hb0[TN], hb1[TN];      // fragments of k-steps 2 and 3 of the previous k-tile
    for (int kt = 0; kt < nkt; ++kt) {
      const char* sa = smem + a_cur * A_BYTES;
      const char* sb = smem + B_BASE + b_cur * B_BYTES;
#ifndef HTRVT_EXP_NOMMA
      if (kt > 0) {          // k-steps 2, 3 of k-tile kt-1, from registers: runs beside the early waves' DMA issue
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha0[i], hb0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha1[i], hb1[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef HTRVT_EXP_NODMA
      if (kt + 1 < nkt) {
        lb.template issue<KMAP>(p, lds0 + B_BASE + b_nxt * B_BYTES, kbeg + (kt + 1) * BK, kend, lw);
        la.issue(p, lds0 + a_nxt * A_BYTES, kbeg + (kt + 1) * BK, kend, lw);
      }
#endif
#ifndef HTRVT_EXP_NOMMA
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, AL>(sa, wm * TM + i, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, BL>(sb, wn * TN + j, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ha0[i] = frag_read<BM, AL>(sa, wm * TM + i, 2, lane);
        ha1[i] = frag_read<BM, AL>(sa, wm * TM + i, 3, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        hb0[j] = frag_read<BN, BL>(sb, wn * TN + j, 2, lane);
        hb1[j] = frag_read<BN, BL>(sb, wn * TN + j, 3, lane);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the held fragments are in registers before this stage is refilled
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      advance();
    }
#ifndef HTRVT_EXP_NOMMA
    if (nkt > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha0[i], hb0[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha1[i], hb1[j], acc[i][j], 0, 0, 0);
    }
#endif
  } else if (SPEC && consumer) {
    // loader-wave kernels (round 5): the MFMA waves run their OWN copy of the loop -- fragment reads and MFMAs between barriers,
    // nothing else -- instead of walking the loaders' branches (gemm_halo_impl.h: -5 ... -9 % per launch on the halo kernels)
    for (int kt = 0; kt < nkt; ++kt) {
      const char* sa = smem + a_cur * A_BYTES;
      const char* sb = smem + B_BASE + b_cur * B_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, AL>(sa, wm * TM + i, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, BL>(sb, wn * TN + j, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();
      advance();
    }
  } else {
  for (int kt = 0; kt < nkt; ++kt) {
    const char* sa = smem + a_cur * A_BYTES;
    const char* sb = smem + B_BASE + b_cur * B_BYTES;
    const int kt_a = kt + AST - 1, kt_b = kt + BST - 1;
#ifndef HTRVT_EXP_NODMA
    if (loader) {   // B first: its (shorter) run-ahead makes it the latency-critical one
      if (kt_b < nkt) lb.template issue<KMAP>(p, lds0 + B_BASE + b_nxt * B_BYTES, kbeg + kt_b * BK, kend, lw);
      if (kt_a < nkt) la.issue(p, lds0 + a_nxt * A_BYTES, kbeg + kt_a * BK, kend, lw);
    }
#endif
#ifdef HTRVT_EXP_NOMMA
    if (false) {
#else
    if (consumer) {
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, AL>(sa, wm * TM + i, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, BL>(sb, wn * TN + j, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    // k-tile kt+1 must have landed before the barrier; what was issued for k-tile kt+2 may stay in flight
    if (NSTAGE == 3 && kt_a < nkt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else if (NSTAGE == 5 && kt_a < nkt)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP_A) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    advance();
  }
  }

  HTRVT_STAMP(3);
  // uniform choice: bf16 C with 16-byte-aligned rows -> staged, vectorised epilogue; float32 C -> direct
  // (256-column tiles serve the float32 split-K weight gradients; their bf16 output, explicit tile selector only,
  //  takes the direct epilogue: the staged one would not leave the loaders' descriptors their SGPRs)
  if (BN <= 192 && !p.c_f32 && ((p.ldc | p.N | coff) & 7) == 0 && ((reinterpret_cast<unsigned long long>(p.C) & 15) == 0)) {
    if constexpr (BN <= 192)
      epilogue_staged<TN, BN, BM, NW_TOTAL, GATHER == 2, GATHER == 1>(acc, p, coff, m0, n0, wm, wn, tile_m, lane, wave, smem, consumer);
  } else
    gemm_epilogue<T, TM, TN, BM / 64, BN, NTH, false>(acc, p, p.C, coff, m0 + wm * TM * 32, n0 + wn * TN * 32, wm, n0, tile_m, lane,
                                               smem, consumer);
  HTRVT_STAMP(6);
#ifdef HTRVT_EXP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HTRVT_STAMP(7);
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    htrvt_dbg[blockIdx.x * 16 + 8] = hw;
  }
#endif
}

template <int BM, int BN, int AL, int BL, int GATHER, int SPEC, int NSTAGE = 2>
__global__ __launch_bounds__(BM * 2 + SPEC * 256) void gemm_dma_kernel(const KParams p) {
  // the parameters are read where they lie, in the kernarg segment (p is the only argument: offset 0), through a
  // constant-address-space reference: scalar loads, never a private copy of the struct
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_dma_body<BM, BN, AL, BL, GATHER, SPEC, NSTAGE>(*kp, (int)blockIdx.x);
}

template <int BM, int BN, int AL, int BL, int GATHER, int SPEC, int NSTAGE = 2>
int launch(const KParams& p, int zdim, hipStream_t st) {
  constexpr int NTH = BM * 2 + SPEC * 256;
  constexpr int smem = NSTAGE == 5 ? 3 * Geo<BM>::BYTES + 2 * Geo<BN>::BYTES : NSTAGE * (Geo<BM>::BYTES + Geo<BN>::BYTES);
  static_assert(smem <= 160 * 1024, "LDS");
  static bool attr_done = false;
  auto kern = gemm_dma_kernel<BM, BN, AL, BL, GATHER, SPEC, NSTAGE>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, zdim);
  if (p.split_k > 1 && ((p.split_k & 7) == 0 || hwgrad_xcd_ranges())) grid = dim3(p.split_k * p.tiles_m * p.tiles_n, 1, 1);   // XCD-grouped K ranges
  hipLaunchKernelGGL(kern, grid, dim3(NTH), smem, st, p);
  set_last_kernel("gemm_dma_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, AL, BL, GATHER, SPEC, NSTAGE);
  const int rc = check_launch("gemm_dma_kernel");
  return rc ? rc : 1;
}

template <int BM, int BN, int SPEC, int NSTAGE = 2>
int dispatch(const HtrvtGemmDesc* d, const KParams& p, int zdim, hipStream_t st) {
  const int al = d->a_layout, bl = d->b_layout, g = d->gather;
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 0) return launch<BM, BN, 0, 0, 0, SPEC, NSTAGE>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 1) return launch<BM, BN, 0, 0, 1, SPEC, NSTAGE>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 2) return launch<BM, BN, 0, 0, 2, SPEC, NSTAGE>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_MNMAJOR && g == 0) return launch<BM, BN, 0, 1, 0, SPEC, NSTAGE>(p, zdim, st);
  if (al == HTRVT_MNMAJOR && bl == HTRVT_MNMAJOR && g == 0) return launch<BM, BN, 1, 1, 0, SPEC, NSTAGE>(p, zdim, st);
  if (al == HTRVT_MNMAJOR && bl == HTRVT_MNMAJOR && g == 3) return launch<BM, BN, 1, 1, 3, SPEC, NSTAGE>(p, zdim, st);
  return 0;
}

}  // namespace
