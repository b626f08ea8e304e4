// conv1x1.hip -- the strided 1x1 "downsample" convolutions of the stem (resnet18.py:59-63), FORWARD, as an HBM-rate
// streaming kernel behind htrvt_gemm (gather = HTRVT_GATHER_CONV_FWD, kh = kw = 1, pad 0).
//
//     y[m][n] = sum_k x[pix(m)][k] * w[n][k],   pix(b, ho, wo) = (b, sh ho, sw wo),   K = Ci in {64 .. 384}
//
// With K = 192 ... 384 such a launch is two HBM passes (read the strided pixels, write y) and 2.5 ... 5 % of the matrix
// peak: on the 256-row-tile GEMM kernels it was all prologue and epilogue (three to six k-tiles per tile, one workgroup
// per CU: 0.46 - 0.60 of the HBM floor, profiles/r04_gemm_table.txt).  Here:
//   * the WEIGHTS live in registers: a wave owns 32 output columns and keeps their K-long rows as MFMA A operands
//     (`n on the MFMA rows`, rows permuted so that a lane's 8 accumulators are 8 consecutive columns of one pixel:
//     one 16-byte store per 16 x 32 block and lane), loaded once per workgroup;
//   * the PIXELS stream through a three-stage LDS ring of 24-KiB stages (64 pixels x 192 channels, 32 x 384, ...) by
//     LDS-DMA with the tile base as the instruction's scalar offset -- per-lane offsets are formed once --, two stages
//     in flight per workgroup, two workgroups per CU, ONE barrier per stage;
//   * stores come straight from the accumulators; the only wait in the loop is a counted vmcnt that lets the previous
//     stages' stores and the next stage's DMA stay outstanding;
//   * the per-M-tile BatchNorm column sums (colstats, the same [ceil(M/256)][2][N] contract as the GEMM kernels) are
//     kept in registers over the four (eight) stages of a 256-row tile and reduced over the 16 pixel lanes by shuffles:
//     every column belongs to exactly one wave, so there is no cross-wave reduction.
#include "gemm_common.h"

using namespace htrvt;

namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr int NSTAGE = 3;       // NWAVE (template): 6 waves = 192 output columns per workgroup, 12 = 384 (N >= 384: the pixels are read once per 384 columns)

// one LDS-DMA piece with a scalar byte offset (the stage's first pixel): LDS[lds_addr + 16 lane] <- buffer[voff + soff]
__device__ __forceinline__ void dma16s(const i32x4_t& rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
      : "memory");
}

__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

// KS = K / 32 k-steps (K = Cpad, a multiple of 64); TR = pixels per stage
template <int KS, int NWAVE>
struct C1 {
  static constexpr int K = 32 * KS, ROWB = 2 * K, TR = K <= 192 ? 64 : 32, RT = TR / 16, SPT = 256 / TR;
  static constexpr int STAGE_BYTES = TR * ROWB;                 // 8 ... 24 KiB
  static constexpr int PIECES = STAGE_BYTES / 1024, NP = (PIECES + NWAVE - 1) / NWAVE;   // DMA pieces per stage / per wave (the surplus: dummies)
  static constexpr int SCRATCH = NSTAGE * STAGE_BYTES, LDS_BYTES = SCRATCH + 1024;
  static_assert(STAGE_BYTES % 1024 == 0 && K <= 384, "stage shape");
  // conflict-free ds_read_b128 of 16 consecutive rows at one k position: rows that are a multiple of 256 B apart need 16
  // different 16-byte slots, 384-byte (128 mod 256) rows alternate bank-line halves and need 8
  static __device__ __forceinline__ int swz(int r) { return (ROWB % 256 == 0) ? (r & 15) : ((r >> 1) & 7); }
};

template <int KS, int NWAVE, class P>
__device__ __forceinline__ void conv1x1_body(const P& p) {
  using G = C1<KS, NWAVE>;
  constexpr int BN = 32 * NWAVE;
  constexpr int K = G::K, TR = G::TR, RT = G::RT, SPT = G::SPT, STAGE_BYTES = G::STAGE_BYTES, NP = G::NP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, jr = lane & 15;
  const int tiles_n = p.tiles_n;
  const int tn = (int)blockIdx.x % tiles_n, wg = (int)blockIdx.x / tiles_n, nwg = (int)gridDim.x / tiles_n;
  const int n0 = tn * BN + wave * 32;                  // this wave's 32 output columns
  const int mtiles = p.tiles_m;
  const int my_tiles = wg < mtiles ? (mtiles - wg + nwg - 1) / nwg : 0;
  const int nst = my_tiles * SPT;
  const unsigned lds0 = lds_addr_of(smem);

  // ---- weights: A operand of v_mfma_f32_16x16x32_bf16, operand row i of column tile t = column 8 (i >> 2) + 4 t + (i & 3) ----
  bf16x8_t wf[2][KS];
  {
    const auto rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, OOB, 0x00020000);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + 8 * (jr >> 2) + 4 * t + (jr & 3);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const unsigned off = col < p.N ? ((unsigned)col * (unsigned)p.ldb + 32 * s + 8 * g) * 2u : OOB;
        wf[t][s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rW, off, 0, 0));
      }
    }
  }

  // ---- pixel loader: lane offsets inside a stage (source chunk swizzled), relative to the stage's first pixel ----
  const unsigned long long ba = (unsigned long long)p.A;
  const i32x4_t rsrcA = i32x4_t{(int)(unsigned)(ba & 0xffffffffull), (int)(unsigned)((ba >> 32) & 0xffffull), (int)OOB, 0x00020000};
  const unsigned pixb = (unsigned)(p.sw * p.Ci) * 2u;              // bytes between consecutive output pixels of a row
  unsigned voff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = (wave + NWAVE * i) * 1024 + lane * 16;
    const int r = q / G::ROWB, c = (q - r * G::ROWB) >> 4;
    const int cs = c ^ G::swz(r);
    voff[i] = (wave + NWAVE * i < G::PIECES && cs * 8 < p.Ci) ? (unsigned)r * pixb + (unsigned)cs * 16u : OOB;      // channels >= Ci (Cpad padding): zeros
  }
  // stage s of this workgroup -> byte offset of its first input pixel (OOB past the end: zero fill)
  auto stage_base = [&](int s) -> unsigned {
    const int mt = wg + (s / SPT) * nwg, sub = s - (s / SPT) * SPT;
    const int m = mt * 256 + sub * TR;
    if (s >= nst || m >= p.M) return OOB;
    const int ho_lin = m / p.Wo, wo = m - ho_lin * p.Wo;           // (b Ho + ho), wo: TR divides Wo (host check)
    const int b = ho_lin / p.Ho, ho = ho_lin - b * p.Ho;
    return (unsigned)(((b * p.Hi + ho * p.sh) * p.Wi + wo * p.sw) * p.Ci) * 2u;
  };
  auto issue = [&](int s, int stage) {
    const unsigned sb = stage_base(s);
#pragma unroll
    for (int i = 0; i < NP; ++i)
      dma16s(rsrcA, __builtin_amdgcn_readfirstlane(wave + NWAVE * i < G::PIECES ? lds0 + stage * STAGE_BYTES + (wave + NWAVE * i) * 1024 : lds0 + G::SCRATCH),
             sb < OOB ? voff[i] : OOB, sb < OOB ? sb : 0u);
  };

  const auto rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, OOB, 0x00020000);
  const unsigned ldc = (unsigned)p.ldc;
  const int ncol = n0 + 8 * g;                          // this lane's 8 consecutive columns
  const bool nok = ncol < p.N;
  float cs1[8], cs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs1[e] = cs2[e] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight fragments
  issue(0, 0);
  issue(1, 1);
  // fragment read: B operand lane (pixel jr, k chunk 4 s + g)
  int stage = 0;
  for (int s = 0; s < nst; ++s) {
    // stage s has landed when everything older than {stores(s-2), DMA(s+1), stores(s-1)} is done
    // (the first two stages have fewer stores behind them: a larger count would not cover their DMA)
    if (s >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 2 * RT) : "memory");
    else if (s == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + RT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    __builtin_amdgcn_s_barrier();
    {
      const int fill = stage >= 1 ? stage - 1 : NSTAGE - 1;       // (stage + 2) % 3: read in iteration s - 1, free behind this barrier
      issue(s + 2, fill);
    }
    const char* sa = smem + stage * STAGE_BYTES;
    f32x4_t acc[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][0] = acc[rt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t fx[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int r = rt * 16 + jr;
        fx[rt] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa + r * G::ROWB + (((4 * ks + g) ^ G::swz(r)) << 4)));
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], fx[rt], acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], fx[rt], acc[rt][1], 0, 0, 0);
      }
    }
    // ---- stores: pixel row m, columns ncol .. ncol + 7 ----
    const int mt = wg + (s / SPT) * nwg, sub = s - (s / SPT) * SPT;
    const int mrow0 = mt * 256 + sub * TR;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int m = mrow0 + rt * 16 + jr;
      float v[8] = {acc[rt][0][0], acc[rt][0][1], acc[rt][0][2], acc[rt][0][3], acc[rt][1][0], acc[rt][1][1], acc[rt][1][2], acc[rt][1][3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) {       // rows >= M hold exact zeros (zero-filled pixels)
        cs1[e] += v[e];
        cs2[e] += v[e] * v[e];
      }
      i32x4_t q;
      q.x = (int)pack_bf16x2(v[0], v[1]); q.y = (int)pack_bf16x2(v[2], v[3]);
      q.z = (int)pack_bf16x2(v[4], v[5]); q.w = (int)pack_bf16x2(v[6], v[7]);
      const unsigned ob = (m < p.M && nok) ? ((unsigned)m * ldc + (unsigned)ncol) * 2u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(q, rC, ob, 0, 0);
    }
    if (sub == SPT - 1) {                 // the 256-row tile is complete: its column sums (workgroup-uniform branch)
      if (p.colstats != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int sft = 1; sft < 16; sft <<= 1) {
            cs1[e] += __shfl_xor(cs1[e], sft, 64);
            cs2[e] += __shfl_xor(cs2[e], sft, 64);
          }
        }
        if (jr == 0 && nok) {
          float* dst = p.colstats + (long long)mt * 2 * p.N + ncol;
          *reinterpret_cast<float4*>(dst) = float4{cs1[0], cs1[1], cs1[2], cs1[3]};
          *reinterpret_cast<float4*>(dst + 4) = float4{cs1[4], cs1[5], cs1[6], cs1[7]};
          *reinterpret_cast<float4*>(dst + p.N) = float4{cs2[0], cs2[1], cs2[2], cs2[3]};
          *reinterpret_cast<float4*>(dst + p.N + 4) = float4{cs2[4], cs2[5], cs2[6], cs2[7]};
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) cs1[e] = cs2[e] = 0.f;
    }
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill pieces issued past the last stage
}

template <int KS, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE, 3) void conv1x1_fwd_kernel(const KParams p) {
  typedef const __attribute__((address_space(4))) KParams KP;
  (void)p;
  KP* kp = (KP*)__builtin_amdgcn_kernarg_segment_ptr();
  conv1x1_body<KS, NWAVE>(*kp);
}

template <int KS, int NWAVE>
int launch_c1(const KParams& p, int nwg, hipStream_t st) {
  static bool attr_done = false;
  auto kern = conv1x1_fwd_kernel<KS, NWAVE>;
  constexpr int smem = C1<KS, NWAVE>::LDS_BYTES;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg * p.tiles_n), dim3(64 * NWAVE), smem, st, p);
  set_last_kernel("conv1x1_fwd_kernel<%d, %d>", KS, NWAVE);
  const int rc = check_launch("conv1x1_fwd_kernel");
  return rc ? rc : 1;
}

}  // namespace

namespace htrvt {

// 1 launched, 0 not served (the GEMM kernels take the launch), < 0 error
int conv1x1_try_launch(const HtrvtGemmDesc* d, KParams& p, hipStream_t st) {
  if (d->gather != HTRVT_GATHER_CONV_FWD || d->dtype != HTRVT_BF16 || d->tile != 0) return 0;
  if (d->kh != 1 || d->kw != 1 || d->ph != 0 || d->pw != 0) return 0;
  if (d->batch > 1 || d->split_k > 1 || d->c_f32 || d->a_layout != HTRVT_KMAJOR || d->b_layout != HTRVT_KMAJOR) return 0;
  // raw output (+ column sums) only: the eval-mode fold (colscale / bias / residual / ReLU) stays on the GEMM kernels' epilogue
  if (d->colscale != nullptr || d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->residual != nullptr || d->alpha != 1.0f) return 0;
  if (d->relu_src != nullptr || d->bnb_partial[0] != nullptr) return 0;
  if (d->K != d->Cpad || d->N != d->Co || (d->Ci & 7) || (d->N & 7) || (d->ldc & 7) || (d->ldb & 7) || d->lda != d->Ci) return 0;
  if ((reinterpret_cast<unsigned long long>(d->C) & 15) || (reinterpret_cast<unsigned long long>(d->B) & 15) || (reinterpret_cast<unsigned long long>(d->A) & 15)) return 0;
  if (d->colstats != nullptr && (reinterpret_cast<unsigned long long>(d->colstats) & 15)) return 0;
  const int ks = d->Cpad / 32;
  if (ks != 2 && ks != 4 && ks != 6 && ks != 8 && ks != 12) return 0;
  const int tr = d->Cpad <= 192 ? 64 : 32;
  if (d->Wo % tr || d->M != d->nB * d->Ho * d->Wo || d->M % tr) return 0;       // a stage = pixels of ONE output row
  const long long lim = (1ll << 31) - 64;
  if ((long long)d->nB * d->Hi * d->Wi * d->Ci * 2 >= lim || (long long)d->M * d->ldc * 2 >= lim || (long long)d->N * d->ldb * 2 >= lim) return 0;
  const int nwave = d->N > 192 ? 12 : 6;
  const int bn = 32 * nwave;
  p.tiles_m = (d->M + 255) / 256;
  p.tiles_n = (d->N + bn - 1) / bn;
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu < 8) ncu = 256;
  }
  int nwg = (nwave == 6 ? 2 : 1) * ncu / p.tiles_n;   // two 6-wave workgroups or one 12-wave workgroup per CU, over all N tiles
  if (nwg > p.tiles_m) nwg = p.tiles_m;
  if (nwg < 1) nwg = 1;
  if (nwave == 6) {
    switch (ks) {
      case 2: return launch_c1<2, 6>(p, nwg, st);
      case 4: return launch_c1<4, 6>(p, nwg, st);
      case 6: return launch_c1<6, 6>(p, nwg, st);
      case 8: return launch_c1<8, 6>(p, nwg, st);
      default: return launch_c1<12, 6>(p, nwg, st);
    }
  }
  switch (ks) {
    case 2: return launch_c1<2, 12>(p, nwg, st);
    case 4: return launch_c1<4, 12>(p, nwg, st);
    case 6: return launch_c1<6, 12>(p, nwg, st);
    case 8: return launch_c1<8, 12>(p, nwg, st);
    default: return launch_c1<12, 12>(p, nwg, st);
  }
}

}  // namespace htrvt
