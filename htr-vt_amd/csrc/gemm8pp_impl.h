// gemm8pp_impl.h -- PERSISTENT form of the 8-phase bfloat16 GEMM (gemm8p_impl.h) for the Linear layers (round 4).
//
//   C[m][n] = act(alpha * sum_k A[m][k] * B[n][k] + bias[n])      A, B K-major, bf16 C, K a multiple of 128
//
// What the one-tile-per-workgroup kernel loses on the K = 768 shapes is outside its k loop: descriptor set-up and the
// first DMA round trip (~3 400 cycles), the epilogue (VALU + a burst of stores while the matrix pipe idles) and the store
// drain before the next workgroup may start (cycle stamps).  Here ONE workgroup per CU walks its tiles
// and the k loop never stops:
//
//  * the LDS-DMA stream runs straight from the last k-tile of tile t into the first k-tiles of tile t+1 (each loader
//    re-targets itself when its k position wraps): no prologue, no DMA round trip between tiles, three half tiles in
//    flight at every moment of the launch;
//  * the EPILOGUE of tile t is folded into the FIRST k-tile of tile t+1: phase q of that k-tile writes accumulator
//    quadrant q with a zero C operand, so the wave flushes quadrant q (scale, bias, GELU, pack, 16-byte row stores) in
//    the load section of that same phase -- while its SIMD partner of the other wave group multiplies.  The flush
//    issues stores only (stores need no wait); the bias of a tile arrives in LDS through one extra DMA piece per wave
//    issued a whole k-tile ahead of the three half tiles the counted wait leaves in flight, and is read with ds_read:
//    nothing in the loop waits on a vector-memory load, which would drain the in-order vmcnt queue and with it the DMA
//    look-ahead;
//  * the only counted wait stays `s_waitcnt vmcnt(6)` per k-tile; in the flush k-tile it is vmcnt(6 + 4 NQ), NQ = the
//    store instructions of one quadrant flush (they are issued between the DMA pieces and may stay outstanding).
//
// Epilogues served: bias (+ exact-erf GELU with the saved pre-activation), and (round 5) bias + RESIDUAL.  A side input
// per element must never be waited for in a way that drains the in-order vmcnt queue (loads, stores and LDS-DMA count
// together).  The residual of a tile therefore arrives in REGISTERS through inline-asm buffer loads the compiler does
// not see (it would put its own, too strict, waits in front of their uses) -- struct Side below:
//   quadrants 0 and 1 of tile t are requested at the START of tile t's LAST k-tile -- older than that k-tile's eight DMA
//   pieces, so the k-tile's own `vmcnt(6)` retires them: no extra wait;
//   quadrants 2 and 3 are requested into the same registers right behind the flushes of quadrants 0 and 1 (phases 1, 2
//   of the next tile's first k-tile) and awaited two phases later with COUNTED waits (vmcnt(4 + NQ + NL), vmcnt(2 + NQ))
//   that leave every younger DMA piece and store in flight.
// All four quadrants at once would need 48 more registers than the 256 a two-waves-per-SIMD kernel has.  GELU' (a side
// input as well) stays on gemm8p_kernel.
#pragma once
#include "gemm8p_impl.h"

namespace g8 {

template <class C>
struct PCfg {
  static constexpr int BIAS_OFF = C::LDS_BYTES;            // 8 x 1 KiB: one bias slot per wave
  static constexpr int LDS_BYTES = C::LDS_BYTES + 8 * 1024;
};

// workgroup-uniform walk over the tiles: virtual block vb -> (m0, n0), the XCD-chunked order of gemm8p_body (blocks b and
// b + 8 share an XCD; gridDim.x is a multiple of 8, so vb and vb + gridDim.x do too)
struct TileWalk {
  int ntiles, tiles_n, q, r;
  __device__ __forceinline__ bool at(int vb, int& m0, int& n0) const {
    if (vb >= ntiles) return false;
    const int xcd = vb & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int tm = id / tiles_n;
    m0 = tm * 256;
    n0 = (id - tm * tiles_n);
    return true;
  }
};

// A operand (activation rows) of a plain K-major GEMM; re-targets itself to the workgroup's next tile when k wraps
template <class C>
struct PLoadA {
  unsigned off0[2][C::NPW];
  int k0, vb;
  i32x4_t rsrc;

  template <class P>
  __device__ __forceinline__ void target(const P& p, const TileWalk& tw, int wave, int lane) {
    int m0 = 0, n0 = 0;
    const bool live = tw.at(vb, m0, n0);
    const int rl = lane >> 3;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) {
        const int r = (wave + 8 * i) * 8 + rl;
        const int cg = (lane & 7) ^ swz(r);
        const int row = m0 + x * C::HM + r;
        off0[x][i] = (live && row < p.M) ? (unsigned)row * (unsigned)(p.lda * 2) + cg * 16 : OOB;
      }
  }
  template <int X, class P>
  __device__ __forceinline__ void issue(const P& p, const TileWalk& tw, unsigned lds_half, int wave, int lane, int stride) {
#pragma unroll
    for (int i = 0; i < C::NPW; ++i)
      dma16(rsrc, __builtin_amdgcn_readfirstlane(lds_half + (wave + 8 * i) * 1024), off0[X][i] + (unsigned)k0 * 2);
    if constexpr (X == 1) {
      k0 += BK;
      if (k0 >= p.K) {     // workgroup-uniform: this operand's next half tile opens the next tile
        k0 = 0;
        vb += stride;
        target(p, tw, wave, lane);
      }
    }
  }
};

template <class C>
struct PLoadB {
  unsigned off0[2][C::NPW];
  int k0, vb;
  i32x4_t rsrc;

  template <class P>
  __device__ __forceinline__ void target(const P& p, const TileWalk& tw, int wave, int lane) {
    int m0 = 0, nt = 0;
    const bool live = tw.at(vb, m0, nt);
    const int n0 = nt * C::BN;
    const int rl = lane >> 3;
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < C::NPW; ++i) {
        const int pi = wave + 8 * i;
        const int r = pi * 8 + rl;
        const int cg = (lane & 7) ^ swz(r);
        const int col = n0 + y * C::HN + bcol<C>(r < C::HN ? r : 0);
        const bool ok = live && pi < C::B_PIECES && col < p.N;
        off0[y][i] = ok ? (unsigned)col * (unsigned)(p.ldb * 2) + cg * 16 : OOB;
      }
  }
  template <int Y, class P>
  __device__ __forceinline__ void issue(const P& p, const TileWalk& tw, unsigned lds_half, unsigned lds_scratch, int wave, int lane,
                                        int stride) {
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
      const bool dm = (C::B_PIECES < 16) && (wave + 8 * i >= C::B_PIECES);      // wave-uniform
      dma16(rsrc, __builtin_amdgcn_readfirstlane(dm ? lds_scratch : lds_half + (wave + 8 * i) * 1024), off0[Y][i] + (unsigned)k0 * 2);
    }
    if constexpr (Y == 1) {
      k0 += BK;
      if (k0 >= p.K) {
        k0 = 0;
        vb += stride;
        target(p, tw, wave, lane);
      }
    }
  }
};

// The residual of two quadrants in flight / in use (E_RES): q[slot][i] = columns 0-7, q2[slot][i] = columns 8-11 (CW = 12) of row
// tile i.  ON = false: no state, no code (the other epilogues compile exactly as before).
template <class C, int MODE>       // MODE 0: no side input; 1: + residual; 2: * GELU'(saved pre-activation)
struct Side {
  __device__ __forceinline__ void init(const void*) {}
  template <int SL, int X, int Y>
  __device__ __forceinline__ void load(int, int, int, int, unsigned, int, int, int, int, int) {}
  template <int SL, int CNT>
  __device__ __forceinline__ void wait() {}
  template <int SL>
  __device__ __forceinline__ void add(int, float*) {}
};
template <class C, int MODE>
struct SideOn {
  static constexpr int MT = C::MT, CW = 4 * C::NT;
  typedef int i32x2_t __attribute__((ext_vector_type(2)));
  i32x4_t q[2][MT];
  i32x2_t q2[2][MT];
  i32x4_t rsrc;
  __device__ __forceinline__ void init(const void* residual) {
    const unsigned long long ba = (unsigned long long)residual;
    rsrc = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), residual != nullptr ? (int)OOB : 0, 0x00020000};
  }
  // request quadrant (X, Y) of the tile at (tm0, tn0) into slot SL: inline asm, invisible to the compiler's wait counting
  template <int SL, int X, int Y>
  __device__ __forceinline__ void load(int tm0, int tn0, int pM, int pN, unsigned ldc, int wr, int wc, int g, int jr, int) {
    const int nb = tn0 + Y * C::HN + wc * C::SN + CW * g;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = tm0 + X * C::HM + wr * C::SM + 16 * i + jr;
      const unsigned ob = (m < pM && nb < pN) ? ((unsigned)m * ldc + (unsigned)nb) * 2u : OOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q[SL][i]) : "v"(ob), "s"(rsrc) : "memory");
      if constexpr (CW == 12)
        asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:16" : "=v"(q2[SL][i]) : "v"(ob), "s"(rsrc) : "memory");
    }
  }
  // counted wait that the uses of slot SL depend on (the registers pass through the statement)
  template <int SL, int CNT>
  __device__ __forceinline__ void wait() {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (CW == 12)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q[SL][i]), "+v"(q2[SL][i]) : "n"(CNT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[SL][i]) : "n"(CNT) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int SL>
  __device__ __forceinline__ void add(int i, float* v) {
    const i32x4_t a = q[SL][i];
    float x[CW];
    x[0] = __uint_as_float((unsigned)a.x << 16); x[1] = __uint_as_float((unsigned)a.x & 0xffff0000u);
    x[2] = __uint_as_float((unsigned)a.y << 16); x[3] = __uint_as_float((unsigned)a.y & 0xffff0000u);
    x[4] = __uint_as_float((unsigned)a.z << 16); x[5] = __uint_as_float((unsigned)a.z & 0xffff0000u);
    x[6] = __uint_as_float((unsigned)a.w << 16); x[7] = __uint_as_float((unsigned)a.w & 0xffff0000u);
    if constexpr (CW == 12) {
      const i32x2_t b = q2[SL][i];
      x[8] = __uint_as_float((unsigned)b.x << 16); x[9] = __uint_as_float((unsigned)b.x & 0xffff0000u);
      x[10] = __uint_as_float((unsigned)b.y << 16); x[11] = __uint_as_float((unsigned)b.y & 0xffff0000u);
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int e = 0; e < CW; ++e) v[e] += x[e];
    } else {
#pragma unroll
      for (int e = 0; e < CW; e += 2) {
        const f32x2_t gg = gelu_erf_grad_fast2(f32x2_t{x[e], x[e + 1]});
        v[e] *= gg.x;
        v[e + 1] *= gg.y;
      }
    }
  }
};
template <class C>
struct Side<C, 1> : SideOn<C, 1> {};
template <class C>
struct Side<C, 2> : SideOn<C, 2> {};

// k-tile flavours
constexpr int KT_PLAIN = 0;      // accumulate
constexpr int KT_OPEN = 1;       // first k-tile of the workgroup's first tile: zero C operand
constexpr int KT_FLUSH = 2;      // first k-tile of a later tile: zero C operand + flush of the previous tile, one quadrant per phase
constexpr int KT_BIAS = 3;       // second k-tile of a tile: + the DMA piece that fetches the tile's bias into the wave's LDS slot
constexpr int KT_SIDE = 4;       // last k-tile of a tile (E_RES): + the register loads of the tile's residual, quadrants 0 and 1

template <class C, int EPI, class P>
__device__ __forceinline__ void gemm8pp_body(const P& p) {
  static_assert(EPI == 0 || EPI == E_GELU || EPI == E_RES, "persistent kernel: bias / GELU / residual epilogues");
  constexpr bool RES = (EPI & E_RES) != 0;          // a side input per element (struct Side)
  constexpr int MT = C::MT, NT = C::NT, CW = 4 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;
  const int wr = wave / C::WARPS_N, wc = wave - wr * C::WARPS_N;
  const int stride = (int)gridDim.x;
  const int pM = p.M, pN = p.N;

  TileWalk tw;
  tw.ntiles = p.tiles_m * p.tiles_n;
  tw.tiles_n = p.tiles_n;
  tw.q = tw.ntiles >> 3;
  tw.r = tw.ntiles & 7;

  PLoadA<C> la;
  PLoadB<C> lb;
  la.rsrc = make_rsrc(p.A);
  lb.rsrc = make_rsrc(p.B);
  la.k0 = lb.k0 = 0;
  la.vb = lb.vb = (int)blockIdx.x;
  la.target(p, tw, wave, lane);
  lb.target(p, tw, wave, lane);

  f32x4_t acc[2][2][MT][NT];

  const int nkt = p.K / BK;          // even, >= 4 (host check)
  const unsigned lds0 = lds_addr_of(smem);
  constexpr unsigned OA0 = 0, OA1 = C::A_HALF, OB0 = 2 * C::A_HALF, OB1 = 2 * C::A_HALF + C::B_HALF;

  const int fr = lane & 15, fg = lane >> 4;
  const int ra = wr * C::SM + fr, rb = wc * C::SN + fr;
  const unsigned rdA = ra * 128 + ((fg ^ swz(ra)) << 4);
  const unsigned rdB = rb * 128 + ((fg ^ swz(rb)) << 4);

  auto stageA = [&](auto xc, auto bufc) {
    constexpr int X = decltype(xc)::value, BUFI = decltype(bufc)::value;
    la.template issue<X>(p, tw, lds0 + BUFI * C::BUF + (X ? OA1 : OA0), wave, lane, stride);
  };
  auto stageB = [&](auto yc, auto bufc) {
    constexpr int Y = decltype(yc)::value, BUFI = decltype(bufc)::value;
    lb.template issue<Y>(p, tw, lds0 + BUFI * C::BUF + (Y ? OB1 : OB0), lds0 + C::SCRATCH, wave, lane, stride);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- epilogue pieces ----
  const int g = lane >> 4, jr = lane & 15;
  auto mk = [](const void* ptr, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000); };
  const auto rC = mk(p.C, OOB);
  const auto rPre = mk(p.preact, p.preact != nullptr ? OOB : 0u);
  const i32x4_t rBias = [&] {
    const unsigned long long ba = (unsigned long long)p.bias;    // NULL: zero records -> every lane reads 0
    return i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), p.bias != nullptr ? p.N * 4 : 0, 0x00020000};
  }();
  const float alpha = p.alpha;
  const unsigned ldc = (unsigned)p.ldc;
  const char* bias_slot = smem + PCfg<C>::BIAS_OFF + wave * 1024;
  typedef int i32x2_t __attribute__((ext_vector_type(2)));
  constexpr int NQ = MT * (CW == 12 ? 2 : 1) * ((EPI & E_GELU) ? 2 : 1);     // store instructions of one quadrant flush
  constexpr int NL = RES ? MT * (CW == 12 ? 2 : 1) : 0;                      // load instructions of one quadrant's residual
  Side<C, (EPI & E_RES) ? 1 : ((EPI & E_GELUGRAD) ? 2 : 0)> side;
  side.init((EPI & E_GELUGRAD) ? (const void*)p.preact : (const void*)p.residual);

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
  // the bias of tile (.., n0) -> this wave's LDS slot: float f = y*SN + c of the slot is column n0 + y*HN + wc*SN + c
  auto stage_bias = [&](int n0) {
    const int f = 4 * lane;
    const int y = f / C::SN, c = f - y * C::SN;
    const int col = n0 + y * C::HN + wc * C::SN + c;
    const unsigned voff = (f < 2 * C::SN && col < pN) ? (unsigned)col * 4u : OOB;
    dma16(rBias, __builtin_amdgcn_readfirstlane(lds0 + PCfg<C>::BIAS_OFF + wave * 1024), voff);
  };
  // quadrant (X, Y) of the tile at (pm0, pn0): NQ store instructions, no loads
  // (g_, jr_: the lane's column group / row -- parameters so that the trailing flush can derive them afresh instead of
  // keeping loop-invariant copies alive through the whole walk: the 256-column GELU variant sits at the 256-register limit)
  auto flush = [&](auto xc, auto yc, int pm0, int pn0, int g_, int jr_, auto slotc) {
    constexpr int X = decltype(xc)::value, Y = decltype(yc)::value, SL = decltype(slotc)::value;
    float bv[CW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias_slot + (Y * C::SN + CW * g_ + 4 * t) * 4);
      bv[4 * t] = b4.x; bv[4 * t + 1] = b4.y; bv[4 * t + 2] = b4.z; bv[4 * t + 3] = b4.w;
    }
    const int nb = pn0 + Y * C::HN + wc * C::SN + CW * g_;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = pm0 + X * C::HM + wr * C::SM + 16 * i + jr_;
      const unsigned ob = (m < pM && nb < pN) ? ((unsigned)m * ldc + (unsigned)nb) * 2u : OOB;
      float v[CW];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) v[4 * t + rg] = acc[X][Y][i][t][rg] * alpha + bv[4 * t + rg];
      if constexpr (EPI & E_GELU) {
        stbf(rPre, ob, v);       // (a NULL preact: zero-record descriptor, the stores are dropped but still counted)
#pragma unroll
        for (int e = 0; e < CW; e += 2) {
          const unsigned w = pack_bf16x2(v[e], v[e + 1]);
          const f32x2_t gg = gelu_erf_fast2(f32x2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)});
          v[e] = gg.x;
          v[e + 1] = gg.y;
        }
      }
      side.template add<SL>(i, v);       // E_RES: + residual, in float32: the sum is rounded once (as gemm8p_kernel does)
      stbf(rC, ob, v);
    }
  };

  // ---- prologue (once per workgroup): k-tile 0 complete + three half tiles of k-tile 1 ----
  stageB(I0{}, I0{});
  stageA(I0{}, I0{});
  stageB(I1{}, I0{});
  stageA(I1{}, I0{});
  stageB(I0{}, I1{});
  stageA(I0{}, I1{});
  stageB(I1{}, I1{});
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 for the whole launch

  bf16x8_t fa[MT][2], fb0[NT][2], fb1[NT][2];

  auto mma = [&](auto xc, auto yc, auto zc, bf16x8_t (&fbx)[NT][2]) {
    constexpr int X = decltype(xc)::value, Y = decltype(yc)::value;
    constexpr bool ZERO = decltype(zc)::value != 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (ZERO && s == 0)
            acc[X][Y][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbx[j][s], fa[i][s], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          else
            acc[X][Y][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbx[j][s], fa[i][s], acc[X][Y][i][j], 0, 0, 0);
        }
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
#define G8P_MFMA_PHASE(X, Y, Z, FB)              \
  __builtin_amdgcn_s_barrier();                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_setprio(1);                 \
  mma(X, Y, Z, FB);                              \
  __builtin_amdgcn_s_setprio(0);                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();

  // one k-tile = four phases (gemm8p_body's schedule); KIND adds the tile-boundary work to the load sections
  auto ktile = [&](auto bufc, auto kindc, int pm0, int pn0, int n0, int m0 = 0) {
    constexpr int BUFI = decltype(bufc)::value, KIND = decltype(kindc)::value;
    using BX = std::integral_constant<int, BUFI>;
    using BY = std::integral_constant<int, BUFI ^ 1>;
    using Z = std::integral_constant<int, (KIND == KT_OPEN || KIND == KT_FLUSH) ? 1 : 0>;
    const char* base = smem + BUFI * C::BUF;
    // phase 1
    if constexpr (KIND == KT_SIDE) {                    // older than this k-tile's DMA pieces: its phase-4 wait retires them
      side.template load<0, 0, 0>(m0, n0, pM, pN, ldc, wr, wc, g, jr, 0);
      side.template load<1, 0, 1>(m0, n0, pM, pN, ldc, wr, wc, g, jr, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    readB(base + OB0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    readA(base + OA0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KIND == KT_BIAS) stage_bias(n0);      // older than the three half tiles the phase-4 wait leaves in flight
    stageA(I1{}, BY{});
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT * 2) : "memory");
    if constexpr (KIND == KT_FLUSH) {
      __builtin_amdgcn_sched_barrier(0);
      flush(I0{}, I0{}, pm0, pn0, g, jr, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RES) {                              // quadrant (1, 1), flushed in phase 3, into the slot just consumed
        side.template load<0, 1, 1>(pm0, pn0, pM, pN, ldc, wr, wc, g, jr, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    G8P_MFMA_PHASE(I0{}, I0{}, Z{}, fb0)
    // phase 2
    readB(base + OB1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    stageB(I0{}, BX{});
    if constexpr (KIND == KT_FLUSH) {
      __builtin_amdgcn_sched_barrier(0);
      flush(I0{}, I1{}, pm0, pn0, g, jr, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RES) {                              // quadrant (1, 0), flushed in phase 4
        side.template load<1, 1, 0>(pm0, pn0, pM, pN, ldc, wr, wc, g, jr, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    G8P_MFMA_PHASE(I0{}, I1{}, Z{}, fb1)
    // phase 3
    readA(base + OA1);
    __builtin_amdgcn_sched_barrier(0);
    stageA(I0{}, BX{});
    if constexpr (KIND == KT_FLUSH) {
      __builtin_amdgcn_sched_barrier(0);
      // younger than the request of quadrant (1, 1): phase 2's two DMA pieces, NQ stores and NL loads, this phase's two pieces
      side.template wait<0, 4 + NQ + NL>();
      flush(I1{}, I1{}, pm0, pn0, g, jr, I0{});
      __builtin_amdgcn_sched_barrier(0);
    }
    G8P_MFMA_PHASE(I1{}, I1{}, Z{}, fb1)
    // phase 4
    if constexpr (KIND == KT_FLUSH) {
      // younger than the request of quadrant (1, 0): phase 3's two DMA pieces and NQ stores
      side.template wait<1, 2 + NQ>();
      flush(I1{}, I0{}, pm0, pn0, g, jr, I1{});
      __builtin_amdgcn_sched_barrier(0);
    }
    stageB(I1{}, BX{});
    // k-tile kt+1 has landed = everything older than the last three half tiles; in the flush k-tile the four quadrants'
    // stores sit between those pieces and A half 1 of kt+1 (issued in phase 1) and may stay outstanding too
    if constexpr (KIND == KT_FLUSH)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + 4 * NQ) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (KIND == KT_SIDE) {     // the same count again, with the residual registers passing through: defined from here on
      side.template wait<0, 6>();
      side.template wait<1, 6>();
    }
    G8P_MFMA_PHASE(I1{}, I0{}, Z{}, fb0)
  };
  using KP = std::integral_constant<int, KT_PLAIN>;
  using KO = std::integral_constant<int, KT_OPEN>;
  using KF = std::integral_constant<int, KT_FLUSH>;
  using KB = std::integral_constant<int, KT_BIAS>;
  using KS = std::integral_constant<int, KT_SIDE>;

  int pm0 = 0, pn0 = 0;
  bool first = true;
  for (int vb = (int)blockIdx.x; vb < tw.ntiles; vb += stride) {
    int m0, nt;
    tw.at(vb, m0, nt);
    const int n0 = nt * C::BN;
    if (first)
      ktile(I0{}, KO{}, 0, 0, n0);
    else
      ktile(I0{}, KF{}, pm0, pn0, n0);
    ktile(I1{}, KB{}, 0, 0, n0);
    if constexpr (RES) {
      for (int kt = 2; kt < nkt - 2; kt += 2) {
        ktile(I0{}, KP{}, 0, 0, n0);
        ktile(I1{}, KP{}, 0, 0, n0);
      }
      ktile(I0{}, KP{}, 0, 0, n0);
      ktile(I1{}, KS{}, 0, 0, n0, m0);      // the tile's last k-tile: + the residual of its quadrants 0 and 1
    } else {
      for (int kt = 2; kt < nkt; kt += 2) {
        ktile(I0{}, KP{}, 0, 0, n0);
        ktile(I1{}, KP{}, 0, 0, n0);
      }
    }
    pm0 = m0;
    pn0 = n0;
    first = false;
  }
#undef G8P_MFMA_PHASE
  if (grp == 0) __builtin_amdgcn_s_barrier();   // group 0 joins group 1's last barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill pieces issued past the last tile
  __builtin_amdgcn_sched_barrier(0);
  // the workgroup's last tile: nothing left to hide it under
  int lane_f = lane;
  asm volatile("" : "+v"(lane_f));
  const int g_f = lane_f >> 4, jr_f = lane_f & 15;
  flush(I0{}, I0{}, pm0, pn0, g_f, jr_f, I0{});
  flush(I0{}, I1{}, pm0, pn0, g_f, jr_f, I1{});
  if constexpr (RES) {
    __builtin_amdgcn_sched_barrier(0);
    side.template load<0, 1, 1>(pm0, pn0, pM, pN, ldc, wr, wc, g_f, jr_f, 0);
    side.template load<1, 1, 0>(pm0, pn0, pM, pN, ldc, wr, wc, g_f, jr_f, 0);
    side.template wait<0, 0>();
    side.template wait<1, 0>();
  }
  flush(I1{}, I1{}, pm0, pn0, g_f, jr_f, I0{});
  flush(I1{}, I0{}, pm0, pn0, g_f, jr_f, I1{});
}

template <class C, int EPI>
__global__ __launch_bounds__(512) void gemm8pp_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  gemm8pp_body<C, EPI>(*kp);
}

// grid: one workgroup per CU (a multiple of 8, at most the tile count)
template <class C, int EPI>
int launch_persistent(const KParams& p, int nwg, hipStream_t st) {
  static_assert(PCfg<C>::LDS_BYTES <= 160 * 1024, "LDS");
  static bool attr_done = false;
  auto kern = gemm8pp_kernel<C, EPI>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PCfg<C>::LDS_BYTES);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", PCfg<C>::LDS_BYTES, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), PCfg<C>::LDS_BYTES, st, p);
  set_last_kernel("gemm8pp_kernel<Cfg<%d, %d, %d>, %d>", C::BN, C::WARPS_M, C::WARPS_N, EPI);
  const int rc = check_launch("gemm8pp_kernel");
  return rc ? rc : 1;
}

}  // namespace g8
