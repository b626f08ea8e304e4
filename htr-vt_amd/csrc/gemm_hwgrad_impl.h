// gemm_hwgrad_impl.h -- weight gradient of the 3x3, stride-1, pad-1 convolutions (resnet18.py:26-31 backward, nine of the
// twelve 3x3 convolutions of the stem) with a HALO-STAGED x operand.
//
//   dW[dy][dx][ci][co] = sum over pixels (b, h, w) of  x[b, h + dy - 1, w + dx - 1, ci] * dY[b, h, w, co]
//
// As a GEMM (gemm_dma_impl.h, GATHER 3) this is M = taps * Cpad rows (tap, ci), N = Co, K = pixels with both operands
// MN-major, and every k-tile of 64 pixels stages a 64 x 256 x-tile in which neighbouring taps hold the SAME pixels shifted
// by one: 56 KB of LDS-DMA per 256 x 192 x 64 MACs, on a path whose per-CU rate (~45 GB/s with everything else going on)
// bounds the loop -- ablation in DESIGN.md 5: MFMA side alone 0.52 ms, DMA alone 0.57 ms, together 0.80 ms.
// Here a workgroup owns ONE kernel row dy, CC input channels, BN output channels and computes all three taps dx of that
// row: per k-tile (64 consecutive pixels of one image row; host check W % 64 == 0) it stages
//   X : the 66 pixels w0-1 .. w0+64 of image row h + dy - 1, CC channels   ([pixel][channel], 8.4 / 16.9 KB)
//   DY: the 64 pixels x BN channels of the output gradient                 ([pixel][channel], 24.6 KB at BN = 192)
// and the MFMA waves of tap dx read their A fragments dx pixel-rows further down the X tile.  Output tile (3 * CC) x BN:
// 41 KB per 384 x 192 x 64 MACs at CC = 128 (-52 % operand bytes per FLOP), 33 KB per 192 x 192 x 64 at CC = 64 (-22 %).
// The smaller stages leave room for THREE of them: the DMA of k-tile t+2 is issued while t is multiplied and the wait before
// the barrier is a counted vmcnt that leaves it in flight.
// 12 waves (6 along M x 2 along N, 64 x 96 or 32 x 96 outputs each, v_mfma_f32_32x32x16_bf16, operands by transposed LDS
// reads), every wave issues its share of the DMA; split-K over pixel ranges with the float32 epilogue (atomics or
// per-range slabs) of the generic kernel, XCD-grouped like there.
#pragma once
#include "gemm_dma_impl.h"

namespace {

template <int CC, int BN>
struct HwGeo {
  static constexpr int NW = 12, NTH = NW * 64;
  static constexpr int BM = 3 * CC;                       // output rows of a tile: (dx, channel)
  static constexpr int XROWB = CC * 2, XCPR = CC / 8;     // X tile row: bytes, 16-byte chunks
  static constexpr int XRPP = 1024 / XROWB;               // pixel rows per DMA piece (4 at CC = 128, 8 at CC = 64)
  static constexpr int XPIECES = (66 + XRPP - 1) / XRPP;  // 17 / 9
  static constexpr int XBYTES = XPIECES * 1024;
  static constexpr int YBYTES = Geo<BN, NW>::BYTES, YPIECES = YBYTES / 1024;
  static constexpr int STAGE = XBYTES + YBYTES;
  static constexpr int NSTAGE = 3;
  static constexpr int LDS_BYTES = NSTAGE * STAGE;
  static constexpr int TM = CC / 64, TN = BN / 64;        // 32x32 tiles per wave: rows (CC / 2 per wave -> 64 or 32), columns BN / 2
  static constexpr int WROWS = CC / 2;                    // rows per wave
  static_assert(CC == 64 || CC == 128, "channel chunk");
  static constexpr int NWY = YPIECES % NW == 0 ? NW : 8;   // waves that issue dY pieces (24 pieces: all 12; 16 pieces: 8)
  static_assert(YPIECES % NWY == 0, "dY pieces divide over the issuing waves");
  static constexpr int NPY = YPIECES / NWY;
  static constexpr int NPX_MAX = (XPIECES + NW - 1) / NW; // waves 0 .. XPIECES - NW*(NPX_MAX-1) - 1 issue NPX_MAX pieces, the rest one fewer
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  // rotation (16-byte chunks) of X pixel row r: the four rows a transposed read of a half-wave touches must land on four
  // different 64-byte bank groups: 256-byte rows -> 4 * (r & 3); 128-byte rows (two per bank line) -> 4 * ((r >> 1) & 1)
  static __device__ __forceinline__ int xrot(int r) { return CC == 128 ? 4 * (r & 3) : 4 * ((r >> 1) & 1); }
};

// PAIR (CC = 128 only): the 128-channel X tile is TWO independent 64-channel units (kernel row, 64-channel chunk) side by
// side -- channels 0-63 of the tile are unit 2 tm, channels 64-127 unit 2 tm + 1, each read from its own image row.  For a
// padded channel count that is a multiple of 64 but not of 128 (layer 1: 192 = three chunks x three kernel rows = nine
// units -> five workgroups) this keeps the 384 x 192 output tile and its 41 KB per k-tile instead of falling back to
// 192 x 192 tiles at 33 KB (CC = 64): -37 % operand bytes per FLOP for nine tenths of the MFMA work.
template <int CC, int BN, bool PAIR, class P>
__device__ __forceinline__ void gemm_hwgrad_body(const P& p, const int block_x) {
  using H = HwGeo<CC, BN>;
  using T = bf16_t;
  constexpr int TM = H::TM, TN = H::TN, NW = H::NW;
  static_assert(!PAIR || CC == 128, "paired units are 64 channels wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int NCC = PAIR ? p.Cpad / 64 : p.Cpad / CC;  // chunks per kernel row (PAIR: of 64 channels)
  const int nunits = 3 * NCC;                        // (kernel row, channel chunk)
  const int tiles_m = PAIR ? (nunits + 1) / 2 : nunits;
  const int ntiles = tiles_m * p.tiles_n;
  int id = block_x, z = blockIdx.z;
  if (p.split_k > 1 && (p.split_k & 7) == 0) {       // all tiles of one pixel range on one XCD (they read the same x / dY rows)
    const int chunk = block_x / (8 * ntiles), r = block_x - chunk * 8 * ntiles;
    z = chunk * 8 + (r & 7);
    id = r >> 3;
  } else if (p.split_k > 1 && gridDim.z == 1) {
    xcd_range_map(block_x, (int)gridDim.x, ntiles, z, id);
  }
  const int tm = id / p.tiles_n, tile_n = id - tm * p.tiles_n;
  // unit u of this workgroup (PAIR: two, else one): kernel row dyu[u], first channel ciu[u]; a missing second unit reads zeros
  int dyu[2], ciu[2];
  bool uok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = PAIR ? 2 * tm + u : tm;
    uok[u] = t < nunits;
    const int tt = uok[u] ? t : 0;
    dyu[u] = tt / NCC;
    ciu[u] = (tt - dyu[u] * NCC) * (PAIR ? 64 : CC);
  }
  const int n0 = tile_n * BN;
  int kbeg = 0, kend = p.K;
  long long coff = 0;
  if (p.split_k > 1) {
    kbeg = z * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
    coff = (long long)z * p.slab_stride;
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;           // 6 x 2
  const int dx = wm / (CC / 64 * 1 + (CC == 64 ? 1 : 0));   // CC = 128: two waves per tap; CC = 64: two waves per tap as well (32 rows each)
  const int crow = (wm & 1) * H::WROWS;              // first channel of this wave inside the chunk

  // ---- DMA bookkeeping ----
  DmaLoader<BN, HTRVT_MNMAJOR, 0, H::NWY> ly;        // dY: plain MN-major rows (k = pixel)
  const bool yload = wave < H::NWY;                  // wave-uniform
  ly.init(p, p.B, p.ldb, n0, p.N, yload ? wave : 0, lane);
  const unsigned long long xa = (unsigned long long)p.A;
  const i32x4_t rsrcX = i32x4_t{(int)(unsigned)(xa & 0xffffffffull), (int)(unsigned)((xa >> 32) & 0xffffull), (int)OOB, 0x00020000};
  const int npx = (wave < H::XPIECES - NW * (H::NPX_MAX - 1)) ? H::NPX_MAX : H::NPX_MAX - 1;   // wave-uniform
  const int per_tile = npx + (wave < H::NWY ? H::NPY : 0);      // DMA pieces this wave issues per k-tile
  auto wait_one_tile_in_flight = [&]() {   // everything but this wave's newest k-tile has landed (the count is an immediate)
    switch (per_tile) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    }
  };
  static_assert(H::NPX_MAX + H::NPY <= 5, "vmcnt switch");
  const unsigned lds0 = lds_addr_of(smem);
  const int Hh = p.Hi, Ww = p.Wi;
  // k-tile position of the NEXT tile to issue: flattened pixel kq = (b * H + h) * W + w0, kept as (row = b*H + h, w0)
  int iq_row = kbeg / Ww, iq_w0 = kbeg - iq_row * Ww, iq_k = kbeg;
  auto issue = [&](int stage) {
    const unsigned sbase = lds0 + stage * H::STAGE;
    // pixel row iq_row = (image, output row ho) of dY; the x row of kernel row dyi is ho * sh + dyi - 1 (sh = 1 or 2, W stride 1)
    const int bimg = iq_row / p.Ho, hrow = iq_row - bimg * p.Ho;
    bool rowok[2];
    unsigned gbase[2];
#pragma unroll
    for (int u = 0; u < (PAIR ? 2 : 1); ++u) {
      const int hh = hrow * p.sh + dyu[u] - 1;
      rowok[u] = uok[u] && iq_k < kend && (unsigned)hh < (unsigned)Hh;
      gbase[u] = (unsigned)(((bimg * Hh + hh) * Ww + iq_w0 - 1) * p.Ci + ciu[u]) * 2u;   // pixel w0 - 1 of the source row (may wrap: masked)
    }
#pragma unroll
    for (int i = 0; i < H::NPX_MAX; ++i) {
      if (i < npx) {      // wave-uniform
        const int pi = wave + NW * i;
        const int r = pi * H::XRPP + lane / H::XCPR;           // pixel row of the halo tile
        const int cd = lane % H::XCPR;
        int cs = cd - H::xrot(r);
        cs += cs < 0 ? H::XCPR : 0;
        const int w = iq_w0 - 1 + r;
        const int u = PAIR ? cs >> 3 : 0;                      // which unit this 16-byte chunk belongs to
        const int cu = PAIR ? cs & 7 : cs;                     // ... and its chunk inside the unit
        const bool ro = PAIR ? (u ? rowok[1] : rowok[0]) : rowok[0];
        const int ci = PAIR ? (u ? ciu[1] : ciu[0]) : ciu[0];
        const unsigned gb = PAIR ? (u ? gbase[1] : gbase[0]) : gbase[0];
        const bool v = ro && r < 66 && (unsigned)w < (unsigned)Ww && ci + cu * 8 < p.Ci;
        const unsigned voff = v ? gb + (unsigned)(r * p.Ci + cu * 8) * 2u : OOB;
        dma16(rsrcX, __builtin_amdgcn_readfirstlane(sbase + pi * 1024), voff);
      }
    }
    if (yload) ly.template issue<false>(p, sbase + H::XBYTES, iq_k, kend, wave);
    iq_k += BK;
    iq_w0 += BK;
    if (iq_w0 >= Ww) {
      iq_w0 = 0;
      ++iq_row;
    }
  };

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (kend - kbeg + BK - 1) / BK;
  issue(0);
  issue(1);
  wait_one_tile_in_flight();      // tile 0 landed, tile 1 may fly
  __builtin_amdgcn_s_barrier();

  // transposed A fragment: rows = channels crow + 32 i + (lane & 31) of tap dx, k = pixels 16 s + ... of the k-tile ->
  // X tile pixel row (k + dx), read by two ds_read_b64_tr_b16 (4 pixel rows each)
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hq = g >> 1;
  auto xfrag = [&](const char* xs, int i, int s) {
    const int r0 = 16 * s + 8 * hq + q + dx;                 // pixel row of the first read; the second is 4 further
    const int cb = (crow + 32 * i) / 8 + 2 * (g & 1) + (pp >> 1);
    int c0 = cb + H::xrot(r0), c1 = cb + H::xrot(r0 + 4);
    c0 -= c0 >= H::XCPR ? H::XCPR : 0;
    c1 -= c1 >= H::XCPR ? H::XCPR : 0;
    typedef __attribute__((address_space(3))) s16x4_t* lptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + r0 * H::XROWB + c0 * 16 + (pp & 1) * 8));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + (r0 + 4) * H::XROWB + c1 * 16 + (pp & 1) * 8));
    const s16x8_t r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };

  // ---- main loop, ONE barrier per k-tile, in the MIDDLE of it ----
  // The barrier used to sit between k-tiles: every wave arrived with its last MFMAs issued, and all twelve then started the
  // next k-tile by requesting fragments with nothing to multiply -- LDS latency plus the issue time of ~130 transposed reads per
  // k-tile boundary with the matrix pipes idle (~10 % of a k-tile).  Now the barrier of k-tile t sits between its k-steps 1 and 2:
  //   * before it a wave waits for ITS pieces of tile t + 1 (issued behind the barrier of tile t - 1: a whole k-tile ago), so
  //     behind it tile t + 1 is complete -- the next k-tile starts without a barrier, in straight-line code that the compiler
  //     software-pipelines like any two k-steps (the reads of (t + 1, step 0) sit between the MFMAs of (t, step 3));
  //   * behind it every wave is past the first half of tile t, so nobody reads tile t - 1 any more: its stage takes the DMA
  //     of tile t + 2;
  //   * the fragments of step 2 are requested BEFORE the wait and the barrier: they are there when the barrier opens.
  auto kstep_load = [&](const char* xs, const char* ys, int s, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = xfrag(xs, i, s);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, HTRVT_MNMAJOR>(ys, wn * TN + j, s, lane);
  };
  auto kstep_mma = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  auto kstep = [&](const char* xs, const char* ys, int s) {
    bf16x8_t fa[TM], fb[TN];
    kstep_load(xs, ys, s, fa, fb);
    kstep_mma(fa, fb);
  };
  bf16x8_t fa2[TM], fb2[TN];          // step 2 of the current k-tile, requested ahead of the barrier
  if (nkt > 0) {                       // first half of tile 0 (landed: prologue barrier)
    kstep(smem, smem + H::XBYTES, 0);
    kstep(smem, smem + H::XBYTES, 1);
    kstep_load(smem, smem + H::XBYTES, 2, fa2, fb2);
  }
  int cur = 0, nxt = 2;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* xs = smem + cur * H::STAGE;
    const char* ys = xs + H::XBYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt + 1
    __builtin_amdgcn_s_barrier();
    issue(nxt);            // k-tile kt + 2 (zero fill past the range) into the stage of tile kt - 1
    kstep_mma(fa2, fb2);
    kstep(xs, ys, 3);
    cur = cur == 2 ? 0 : cur + 1;
    nxt = nxt == 2 ? 0 : nxt + 1;
    if (kt + 1 < nkt) {    // first half of the next tile: complete since the barrier above
      const char* xn = smem + cur * H::STAGE;
      const char* yn = xn + H::XBYTES;
      kstep(xn, yn, 0);
      kstep(xn, yn, 1);
      kstep_load(xn, yn, 2, fa2, fb2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // rows of the packed weight gradient [tap][Cpad][Co]: m = (dy * 3 + dx) * Cpad + first channel of the unit + channel
  // (PAIR: a wave's 64 rows are one unit -- crow = 0 is unit 0, crow = 64 unit 1; an absent unit stores nothing)
  const int un = PAIR ? (wm & 1) : 0;
  const int mrow0 = (dyu[un] * 3 + dx) * p.Cpad + ciu[un] + (PAIR ? 0 : crow);
  gemm_epilogue<T, TM, TN, 1, BN, H::NTH, false>(acc, p, p.C, coff, mrow0, n0 + wn * TN * 32, 0, n0, 0, lane, smem, uok[un]);
}

// ---------------------------------------------------------------------------------------------
// Round 5: the same kernel on v_mfma_f32_16x16x32_bf16 with 96-channel units (gemm_hwgrad16_kernel<96, BN>).
//
// Layer 1 (192 channels) does not divide into the 128-channel chunks above: nine (kernel row, 64-channel chunk) units made
// five PAIRED 384 x 192 tiles with the last one half empty (-10 %) and 5 x 48 K ranges filled 240 of 256 CUs (-6 %)
// (1 019 TFLOP/s against 1 201-1 246 at layers 2-3).  With 96-channel chunks the 192 channels
// are 3 x 2 = SIX full units of 288 x 192: 6 x 42 K ranges = 252 workgroups, nothing half empty.  A wave's 48 x 96 block is
// 3 x 6 tiles of 16 x 16 (48 rows have no 32-row tiling): operands by two ds_read_b64_tr_b16 per 16 x 32 fragment -- the lane
// groups of one read now take k-rows 8 g + q, so the LDS images are rotated by 32 bytes per 8 k-rows on top of the 64-byte
// steps (X, 192-byte rows: 2 * ((r >> 3) & 1) chunks; dY, 384-byte rows: 4 * ((k >> 1) & 1) + 2 * ((k >> 3) & 1)) --
// conflict-free by exhaustive check of the bank rule.  72 accumulators + two k-steps of fragments (one held across the
// mid-tile barrier) fit the 168 registers of twelve waves; a 128-channel unit (96 accumulators) would not, so layers 2-3 stay
// on the kernel above.  The chip also holds a higher clock on this MFMA shape (gemm_halo_impl.h).
// ---------------------------------------------------------------------------------------------
// SW = 2 (round 5): the weight gradients of the COLUMN-strided 3x3 convolutions (conv1 of layer2.0 / layer3.0, stride (2,2)).  A k-tile
// is still 64 output pixels wo0 .. wo0 + 63 of one output row; kernel columns 0 and 2 contract them against the ODD input pixels
// 2 wo0 - 1 + 2 r (column 2 one row further down) and column 1 against the EVEN pixels 2 wo0 + 2 r: the x stage holds an odd image
// (LDS rows 0 .. 64) and an even image (rows 80 .. 143: a multiple of 16 rows further down, so the bank pattern of the transposed
// reads is the one checked for SW = 1) -- 129 pixel rows per k-tile instead of the 3 x 64 of the generic gather, which decoded a
// pixel per row and tap (layer2.0 / 3.0: 0.41 / 0.37 ms = 850 / 950 TFLOP/s on gemm_dma_kernel<.., 3, ..>).
template <int CC, int BN, int SW = 1>
struct Hw16Geo {
  static constexpr int NW = 12, NTH = NW * 64;
  static constexpr int XROWB = CC * 2, XCPR = CC / 8;
  static constexpr int EVEN0 = 80;                             // first LDS row of the even image (SW = 2)
  static constexpr int XROWS = SW == 1 ? 66 : EVEN0 + 64;
  static constexpr int XPIECES = (XROWS * XROWB + 1023) / 1024;   // 13 at CC = 96 (SW = 2: 27)
  static constexpr int XBYTES = XPIECES * 1024;
  static constexpr int YROWB = BN * 2, YCPR = BN / 8;
  static constexpr int YBYTES = 64 * YROWB, YPIECES = YBYTES / 1024;
  static constexpr int STAGE = XBYTES + YBYTES, NSTAGE = 3, LDS_BYTES = NSTAGE * STAGE;
  static constexpr int RT = CC / 32, CT = BN / 32;             // 16 x 16 tiles per wave: rows (CC / 2 channels), columns (BN / 2)
  static constexpr int NPX = (XPIECES + NW - 1) / NW, NPY = YPIECES / NW;
  static_assert(CC == 96 && BN == 192 && (SW == 1 || SW == 2) && YPIECES % NW == 0 && LDS_BYTES <= 160 * 1024, "geometry");
  // one k-step's fragments held across the mid-tile barrier (gemm_hwgrad_body)
  static constexpr bool HOLD = CC == 96;
  static __device__ __forceinline__ int xrot(int r) { return CC == 96 ? 2 * ((r >> 3) & 1) : 4 * (r & 3) + 2 * ((r >> 3) & 1); }
  static __device__ __forceinline__ int yrot(int k) { return BN == 192 ? 4 * ((k >> 1) & 1) + 2 * ((k >> 3) & 1) : 4 * (k & 3) + 2 * ((k >> 3) & 1); }
};

typedef float f32x4w_t __attribute__((ext_vector_type(4)));

template <int CC, int BN, int SW, class P>
__device__ __forceinline__ void gemm_hwgrad16_body(const P& p, const int block_x) {
  using H = Hw16Geo<CC, BN, SW>;
  constexpr int NW = H::NW, RT = H::RT, CT = H::CT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int NCC = p.Cpad / CC;                       // chunks per kernel row
  const int nunits = 3 * NCC;
  const int ntiles = nunits * p.tiles_n;
  int id = block_x, z = blockIdx.z;
  if (p.split_k > 1 && (p.split_k & 7) == 0) {       // all tiles of one pixel range on one XCD (they read the same x / dY rows)
    const int chunk = block_x / (8 * ntiles), r = block_x - chunk * 8 * ntiles;
    z = chunk * 8 + (r & 7);
    id = r >> 3;
  } else if (p.split_k > 1 && gridDim.z == 1) {
    xcd_range_map(block_x, (int)gridDim.x, ntiles, z, id);
  }
  const int tm = id / p.tiles_n, tile_n = id - tm * p.tiles_n;
  const int dyu = tm / NCC, ciu = (tm - dyu * NCC) * CC;
  const int n0 = tile_n * BN;
  int kbeg = 0, kend = p.K;
  long long coff = 0;
  if (p.split_k > 1) {
    kbeg = z * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
    coff = (long long)z * p.slab_stride;
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;           // 6 x 2
  const int dx = wm >> 1;                            // two waves per tap, CC / 2 channels each
  const int crow = (wm & 1) * (CC / 2);

  // ---- DMA bookkeeping: per piece (row / k-row, source chunk) of this lane, formed once ----
  const unsigned long long xa = (unsigned long long)p.A, ya = (unsigned long long)p.B;
  const i32x4_t rsrcX = i32x4_t{(int)(unsigned)(xa & 0xffffffffull), (int)(unsigned)((xa >> 32) & 0xffffull), (int)OOB, 0x00020000};
  const i32x4_t rsrcY = i32x4_t{(int)(unsigned)(ya & 0xffffffffull), (int)(unsigned)((ya >> 32) & 0xffffull), (int)OOB, 0x00020000};
  int xr[H::NPX], xc[H::NPX];      // xr: source pixel of this lane's LDS row, relative to the k-tile's first (SW = 1: w0 - 1; SW = 2: 2 w0)
  bool xok[H::NPX];
#pragma unroll
  for (int i = 0; i < H::NPX; ++i) {
    const int pi = wave + NW * i;
    const int q = pi * 1024 + lane * 16;
    const int r = q / H::XROWB, cd = (q - r * H::XROWB) >> 4;
    int cs = cd - H::xrot(r);
    cs += cs < 0 ? H::XCPR : 0;
    xr[i] = SW == 1 ? r : (r < H::EVEN0 ? 2 * r - 1 : 2 * (r - H::EVEN0));
    xc[i] = cs;
    xok[i] = pi < H::XPIECES && (SW == 1 ? r < 66 : (r < 65 || (r >= H::EVEN0 && r < H::EVEN0 + 64))) && ciu + cs * 8 < p.Ci;
  }
  unsigned yoff[H::NPY];
  int ykr[H::NPY];
#pragma unroll
  for (int i = 0; i < H::NPY; ++i) {
    const int sidx = (wave + NW * i) * 64 + lane;
    const int kr = sidx / H::YCPR, cl = sidx - kr * H::YCPR;
    int cg = cl - H::yrot(kr);
    cg += cg < 0 ? H::YCPR : 0;
    const int col = n0 + cg * 8;
    ykr[i] = kr;
    yoff[i] = col < p.N ? (unsigned)kr * (unsigned)(p.ldb * 2) + (unsigned)col * 2u : OOB;
  }
  const unsigned lds0 = lds_addr_of(smem);
  const int Hh = p.Hi, Ww = p.Wi, Wk = p.Wo;        // a k-tile = 64 pixels of one OUTPUT row (Wo = Wi at SW = 1)
  int iq_row = kbeg / Wk, iq_w0 = kbeg - iq_row * Wk, iq_k = kbeg;
  auto issue = [&](int stage) {
    const unsigned sbase = lds0 + stage * H::STAGE;
    const int bimg = iq_row / p.Ho, hrow = iq_row - bimg * p.Ho;
    const int hh = hrow * p.sh + dyu - 1;
    const bool rowok = iq_k < kend && (unsigned)hh < (unsigned)Hh;
    const int wfirst = SW == 1 ? iq_w0 - 1 : 2 * iq_w0;
    const unsigned gbase = (unsigned)(((bimg * Hh + hh) * Ww + wfirst) * p.Ci + ciu) * 2u;   // pixel wfirst of the source row (may wrap: masked)
#pragma unroll
    for (int i = 0; i < H::NPX; ++i) {
      if (wave + NW * i < H::XPIECES) {      // wave-uniform
        const int w = wfirst + xr[i];
        const bool v = rowok && xok[i] && (unsigned)w < (unsigned)Ww;
        const unsigned voff = v ? gbase + (unsigned)(xr[i] * p.Ci + xc[i] * 8) * 2u : OOB;
        dma16(rsrcX, __builtin_amdgcn_readfirstlane(sbase + (wave + NW * i) * 1024), voff);
      }
    }
    const unsigned ybase = (unsigned)iq_k * (unsigned)(p.ldb * 2);
#pragma unroll
    for (int i = 0; i < H::NPY; ++i) {
      const bool v = yoff[i] < OOB && iq_k + ykr[i] < kend;
      dma16(rsrcY, __builtin_amdgcn_readfirstlane(sbase + H::XBYTES + (wave + NW * i) * 1024), v ? ybase + yoff[i] : OOB);
    }
    iq_k += BK;
    iq_w0 += BK;
    if (iq_w0 >= Wk) {
      iq_w0 = 0;
      ++iq_row;
    }
  };

  f32x4w_t acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = f32x4w_t{0.f, 0.f, 0.f, 0.f};

  const int nkt = (kend - kbeg + BK - 1) / BK;
  issue(0);
  issue(1);
  // this wave's pieces of tile 0 have landed, tile 1 may fly (per-wave piece count: NPY + 1 or 2 X pieces)
  if (wave + NW * (H::NPX - 1) < H::XPIECES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NPX + H::NPY) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H::NPX - 1 + H::NPY) : "memory");
  __builtin_amdgcn_s_barrier();

  // 16 x 32 operand fragments by two transposed reads: lane (g, q, pp) supplies k-row 8 g + q (then + 4), columns 4 pp .. 4 pp + 3 of the
  // tile's 16; it receives column (lane & 15), k = 8 g .. 8 g + 7
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  typedef __attribute__((address_space(3))) s16x4_t* lptr;
  auto xfrag = [&](const char* xs, int i, int s) {
    // SW = 2: kernel column 0 / 2 = odd image at row shift 0 / 1, column 1 = even image
    const int r0 = 32 * s + 8 * g + q + (SW == 1 ? dx : (dx == 1 ? H::EVEN0 : (dx >> 1)));
    const int cb = (crow + 16 * i) / 8 + (pp >> 1);
    int c0 = cb + H::xrot(r0), c1 = cb + H::xrot(r0 + 4);
    c0 -= c0 >= H::XCPR ? H::XCPR : 0;
    c1 -= c1 >= H::XCPR ? H::XCPR : 0;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + r0 * H::XROWB + c0 * 16 + (pp & 1) * 8));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xs + (r0 + 4) * H::XROWB + c1 * 16 + (pp & 1) * 8));
    const s16x8_t r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };
  auto yfrag = [&](const char* ys, int j, int s) {
    const int k0 = 32 * s + 8 * g + q;
    const int cb = (wn * (BN / 2) + 16 * j) / 8 + (pp >> 1);
    int c0 = cb + H::yrot(k0), c1 = cb + H::yrot(k0 + 4);
    c0 -= c0 >= H::YCPR ? H::YCPR : 0;
    c1 -= c1 >= H::YCPR ? H::YCPR : 0;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ys + k0 * H::YROWB + c0 * 16 + (pp & 1) * 8));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ys + (k0 + 4) * H::YROWB + c1 * 16 + (pp & 1) * 8));
    const s16x8_t r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return __builtin_bit_cast(bf16x8_t, r);
  };
  auto kstep_load = [&](const char* xs, const char* ys, int s, bf16x8_t (&fa)[RT], bf16x8_t (&fb)[CT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i) fa[i] = xfrag(xs, i, s);
#pragma unroll
    for (int j = 0; j < CT; ++j) fb[j] = yfrag(ys, j, s);
  };
  auto kstep_mma = [&](const bf16x8_t (&fa)[RT], const bf16x8_t (&fb)[CT]) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  if constexpr (H::HOLD) {
  // one barrier per k-tile, between its two k-steps (see gemm_hwgrad_body): the fragments of step 1 are requested before it
  bf16x8_t fa2[RT], fb2[CT];
  if (nkt > 0) {
    bf16x8_t fa[RT], fb[CT];
    kstep_load(smem, smem + H::XBYTES, 0, fa, fb);
    kstep_mma(fa, fb);
    kstep_load(smem, smem + H::XBYTES, 1, fa2, fb2);
  }
  int cur = 0, nxt = 2;
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt + 1
    __builtin_amdgcn_s_barrier();
    issue(nxt);            // k-tile kt + 2 (zero fill past the range) into the stage of tile kt - 1
    kstep_mma(fa2, fb2);
    cur = cur == 2 ? 0 : cur + 1;
    nxt = nxt == 2 ? 0 : nxt + 1;
    if (kt + 1 < nkt) {    // first half of the next tile: complete since the barrier above
      const char* xn = smem + cur * H::STAGE;
      const char* yn = xn + H::XBYTES;
      bf16x8_t fa[RT], fb[CT];
      kstep_load(xn, yn, 0, fa, fb);
      kstep_mma(fa, fb);
      kstep_load(xn, yn, 1, fa2, fb2);
    }
  }
  } else {
  // barrier between k-tiles: tile kt + 1 landed (own pieces waited for) before it, tile kt + 2 issued behind it
  int cur = 0, nxt = 2;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* xs = smem + cur * H::STAGE;
    const char* ys = xs + H::XBYTES;
    if (kt > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    issue(nxt);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8_t fa[RT], fb[CT];
      kstep_load(xs, ys, s2, fa, fb);
      kstep_mma(fa, fb);
    }
    cur = cur == 2 ? 0 : cur + 1;
    nxt = nxt == 2 ? 0 : nxt + 1;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: rows of the packed weight gradient [tap][Cpad][Co]: m = (dy * 3 + dx) * Cpad + first channel of the unit + channel ----
  // acc[i][j][r] = C(channel crow + 16 i + 4 g + r, column wn * BN / 2 + 16 j + (lane & 15)); float32, plain stores into the K range's slab
  // or atomic accumulation (HtrvtGemmDesc.accumulate without splitk_ws)
  float* Cf = reinterpret_cast<float*>(p.C) + coff;
  const int mbase = (dyu * 3 + dx) * p.Cpad + ciu + crow + 4 * g;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int n = n0 + wn * (BN / 2) + 16 * j + (lane & 15);
      if (n < p.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = Cf + (long long)(mbase + 16 * i + r) * p.ldc + n;
          const float v = acc[i][j][r] * p.alpha;
          if (p.accumulate) atomicAdd(dst, v);
          else *dst = v;
        }
      }
    }
}

template <int CC, int BN, int SW = 1>
__global__ __launch_bounds__(768) void gemm_hwgrad16_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_hwgrad16_body<CC, BN, SW>(*kp, (int)blockIdx.x);
}

template <int CC, int BN, int SW = 1>
int launch_hwgrad16(const KParams& p, int zdim, hipStream_t st) {
  using H = Hw16Geo<CC, BN, SW>;
  static bool attr_done = false;
  auto kern = gemm_hwgrad16_kernel<CC, BN, SW>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, H::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", H::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  const int ntiles = 3 * (p.Cpad / CC) * p.tiles_n;
  dim3 grid(ntiles, 1, zdim);
  if (p.split_k > 1 && ((p.split_k & 7) == 0 || hwgrad_xcd_ranges())) grid = dim3(p.split_k * ntiles, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(768), H::LDS_BYTES, st, p);
  if (SW == 1) set_last_kernel("gemm_hwgrad16_kernel<%d, %d>", CC, BN);
  else set_last_kernel("gemm_hwgrad16_kernel<%d, %d, %d>", CC, BN, SW);
  const int rc = check_launch("gemm_hwgrad16_kernel");
  return rc ? rc : 1;
}

template <int CC, int BN, bool PAIR = false>
__global__ __launch_bounds__(768) void gemm_hwgrad_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_hwgrad_body<CC, BN, PAIR>(*kp, (int)blockIdx.x);
}

template <int CC, int BN, bool PAIR = false>
int launch_hwgrad(const KParams& p, int zdim, hipStream_t st) {
  using H = HwGeo<CC, BN>;
  static bool attr_done = false;
  auto kern = gemm_hwgrad_kernel<CC, BN, PAIR>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, H::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", H::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  const int ntiles = (PAIR ? (3 * (p.Cpad / 64) + 1) / 2 : 3 * (p.Cpad / CC)) * p.tiles_n;
  dim3 grid(ntiles, 1, zdim);
  if (p.split_k > 1 && ((p.split_k & 7) == 0 || hwgrad_xcd_ranges())) grid = dim3(p.split_k * ntiles, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(768), H::LDS_BYTES, st, p);
  set_last_kernel(PAIR ? "gemm_hwgrad_kernel<%d, %d, true>" : "gemm_hwgrad_kernel<%d, %d>", CC, BN);
  const int rc = check_launch("gemm_hwgrad_kernel");
  return rc ? rc : 1;
}

}  // namespace
