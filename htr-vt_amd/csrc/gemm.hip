// gemm.hip -- the one MFMA contraction kernel of the HTR-VT hot path (gfx950).
//
//   C[m][n] = alpha * sum_k A(m,k) * B(n,k)   (+bias, GELU, residual, BN column sums)
//
// One template covers
//   * nn.Linear forward / dgrad / wgrad            (reference HTR_VT.py:22,29-37,170, timm Mlp)
//   * attention QK^T, PV and their backward (batched)              (HTR_VT.py:32-36)
//   * 3x3 / 1x1 convolution forward / dgrad / wgrad as implicit GEMM over NHWC
//     (reference resnet18.py:6-7,26-31,59-63): the A (or B) rows are gathered
//     from shifted pixels, out-of-image taps are zero-filled in the loader.
//
// CDNA4 mapping: 256 threads = 4 waves (2x2), each wave owns a (BM/2)x(BN/2)
// block of 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16, or the exact-f32
// v_mfma_f32_32x32x2_f32 for the parity path).  Operand tiles are staged
// global -> registers -> LDS (double buffered, loads for tile t+1 in flight
// while tile t is multiplied).  K-contiguous operands use 128-byte LDS rows with
// a 16-byte-chunk XOR swizzle (conflict-free ds_read_b128); operands whose
// reduction index is strided in memory (dgrad / wgrad / PV) are kept in their
// memory order and read with ds_read_b64_tr_b16 (hardware transpose).
#include "gemm_common.h"

using namespace htrvt;

namespace {

constexpr int NTHREADS = 256;

template <typename T, int ROWS, int LAYOUT>
struct TileGeom {
  static constexpr int SZ = ET<T>::SZ;
  static constexpr int BK = ET<T>::BK;
  static constexpr int STRIDE = (LAYOUT == HTRVT_KMAJOR) ? 128 : (ROWS * SZ + 64);  // LDS bytes per row
  static constexpr int LROWS = (LAYOUT == HTRVT_KMAJOR) ? ROWS : BK;
  static constexpr int BYTES = STRIDE * LROWS;
  static constexpr int NLOAD = ROWS / 32;                                        // 16-B loads / thread / k-tile
  static constexpr int CPR = (LAYOUT == HTRVT_KMAJOR) ? 8 : (ROWS * SZ / 16);    // 16-B chunks per LDS row
};

// ---------------------------------------------------------------------------------------------
// operand loader: global -> registers (issue) -> LDS (commit)
// ROLE: 0 plain, 1 conv-fwd rows, 2 conv-dgrad rows (both K-major A), 3 conv-wgrad k-rows (MN-major B)
// ---------------------------------------------------------------------------------------------
template <typename T, int ROWS, int LAYOUT, int ROLE>
struct Loader {
  using G = TileGeom<T, ROWS, LAYOUT>;
  static constexpr int SZ = G::SZ, CH = ET<T>::CH, NL = G::NLOAD;
  uint4 reg[NL];
  const char* ptr[NL];
  int c0[NL], c1[NL], c2[NL];
  bool ok[NL];
  int ldsoff[NL];
  const char* base;
  long long ld;

  __device__ __forceinline__ void init(const KParams& p, const char* base_, long long ld_, int row0, int rows_total) {
    base = base_;
    ld = ld_;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if constexpr (LAYOUT == HTRVT_KMAJOR) {
        const int rl = (tid >> 3) + 32 * i, chunk = tid & 7;
        const int row = row0 + rl;
        ok[i] = row < rows_total;
        ldsoff[i] = rl * 128 + ((chunk ^ ((rl >> 1) & 7)) << 4);
        c2[i] = chunk * CH;  // k offset of this chunk inside a k-tile
        if constexpr (ROLE == 0) {
          ptr[i] = base + ((long long)row * ld + chunk * CH) * SZ;
          c0[i] = c1[i] = 0;
        } else if constexpr (ROLE == 1) {  // row = output pixel (b, ho, wo)
          const int hw = p.Ho * p.Wo;
          const int b = row / hw, r = row - b * hw;
          const int ho = r / p.Wo, wo = r - ho * p.Wo;
          ptr[i] = base + (long long)b * p.Hi * p.Wi * p.Ci * SZ;
          c0[i] = ho * p.sh - p.ph;
          c1[i] = wo * p.sw - p.pw;
        } else {  // ROLE 2: row = input pixel (b, hi, wi)
          const int hw = p.Hi * p.Wi;
          const int b = row / hw, r = row - b * hw;
          const int hi = r / p.Wi, wi = r - hi * p.Wi;
          ptr[i] = base + (long long)b * p.Ho * p.Wo * p.Co * SZ;
          c0[i] = hi + p.ph;
          c1[i] = wi + p.pw;
        }
      } else {
        const int idx = tid + NTHREADS * i;
        const int krow = idx / G::CPR, chunk = idx - krow * G::CPR;
        const int col = row0 + chunk * CH;
        ldsoff[i] = krow * G::STRIDE + chunk * 16;
        c2[i] = krow;
        if constexpr (ROLE == 0) {
          ok[i] = col < rows_total;
          ptr[i] = base + (long long)col * SZ;
          c0[i] = c1[i] = 0;
        } else {  // ROLE 3: col = tap*Cpad + ci of the packed weight-gradient
          const int tap = col / p.Cpad, ci = col - tap * p.Cpad;
          const int dy = tap / p.kw, dx = tap - dy * p.kw;
          ok[i] = (col < rows_total) && (ci < p.Ci);
          ptr[i] = base + (long long)ci * SZ;
          c0[i] = dy - p.ph;
          c1[i] = dx - p.pw;
        }
      }
    }
  }

  // k0: first k of the tile; kend: exclusive end of this block's k range
  __device__ __forceinline__ void issue(const KParams& p, int k0, int kend) {
    int tap_dy = 0, tap_dx = 0, cbase = k0;
    if constexpr (ROLE == 1 || ROLE == 2) {
      const int tap = k0 / p.Cpad;
      cbase = k0 - tap * p.Cpad;
      tap_dy = tap / p.kw;
      tap_dx = tap - tap_dy * p.kw;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bool v = ok[i];
      const char* src = ptr[i];
      if constexpr (LAYOUT == HTRVT_KMAJOR) {
        if constexpr (ROLE == 0) {
          v = v && (k0 + c2[i] < kend);
          src += (long long)k0 * SZ;
        } else if constexpr (ROLE == 1) {
          const int hi = c0[i] + tap_dy, wi = c1[i] + tap_dx, c = cbase + c2[i];
          v = v && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi) && (c < p.Ci);
          src += ((long long)(hi * p.Wi + wi) * p.Ci + c) * SZ;
        } else {
          const int th = c0[i] - tap_dy, tw = c1[i] - tap_dx, c = cbase + c2[i];
          const int ho = th >> (p.sh - 1), wo = tw >> (p.sw - 1);
          v = v && (th >= 0) && (tw >= 0) && ((th & (p.sh - 1)) == 0) && ((tw & (p.sw - 1)) == 0) && (ho < p.Ho) &&
              (wo < p.Wo) && (c < p.Co);
          src += ((long long)(ho * p.Wo + wo) * p.Co + c) * SZ;
        }
      } else {
        const int k = k0 + c2[i];
        v = v && (k < kend);
        if constexpr (ROLE == 0) {
          src += (long long)k * ld * SZ;
        } else {  // k = output pixel
          const int hw = p.Ho * p.Wo;
          const int b = k / hw, r = k - b * hw;
          const int ho = r / p.Wo, wo = r - ho * p.Wo;
          const int hi = ho * p.sh + c0[i], wi = wo * p.sw + c1[i];
          v = v && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          src += ((long long)(b * p.Hi + hi) * p.Wi + wi) * p.Ci * SZ;
        }
      }
      reg[i] = v ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
    }
  }

  __device__ __forceinline__ void commit(char* lds) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<uint4*>(lds + ldsoff[i]) = reg[i];
  }
};

// ---------------------------------------------------------------------------------------------
// fragment reads
// ---------------------------------------------------------------------------------------------
template <int ROWS, int LAYOUT>
__device__ __forceinline__ bf16x8_t frag_bf16(const char* lds, int rb, int s, int lane) {
  using G = TileGeom<bf16_t, ROWS, LAYOUT>;
  if constexpr (LAYOUT == HTRVT_KMAJOR) {
    const int row = rb * 32 + (lane & 31);
    const int chunk = 2 * s + (lane >> 5);
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    return __builtin_bit_cast(bf16x8_t, v);
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
    const int col = rb * 32 + 16 * (g & 1) + 4 * pp;
    const int krow = 16 * s + 8 * h + q;
    const char* a0 = lds + krow * G::STRIDE + col * 2;
    typedef __attribute__((address_space(3))) s16x4_t* lptr;
    const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0));
    const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(a0 + 4 * G::STRIDE));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t r = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    return __builtin_bit_cast(bf16x8_t, r);
  }
}

// f32: one "u" group = 4 MFMA 32x32x2 steps; step (u,j) contracts k_eff = 8u + 4h + j (h = lane>>5)
template <int ROWS, int LAYOUT>
__device__ __forceinline__ float4 frag_f32(const char* lds, int rb, int u, int lane) {
  using G = TileGeom<float, ROWS, LAYOUT>;
  const int h = lane >> 5;
  if constexpr (LAYOUT == HTRVT_KMAJOR) {
    const int row = rb * 32 + (lane & 31);
    const int chunk = 2 * u + h;
    return *reinterpret_cast<const float4*>(lds + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  } else {
    const char* a = lds + (8 * u + 4 * h) * G::STRIDE + (rb * 32 + (lane & 31)) * 4;
    float4 r;
    r.x = *reinterpret_cast<const float*>(a);
    r.y = *reinterpret_cast<const float*>(a + G::STRIDE);
    r.z = *reinterpret_cast<const float*>(a + 2 * G::STRIDE);
    r.w = *reinterpret_cast<const float*>(a + 3 * G::STRIDE);
    return r;
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int AL, int BL, int GATHER>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const KParams p) {
  using GA = TileGeom<T, BM, AL>;
  using GB = TileGeom<T, BN, BL>;
  constexpr int BK = ET<T>::BK;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int STAGE = GA::BYTES + GB::BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- block -> tile (XCD-aware: blocks that share an XCD get neighbouring tiles) ----
  const int ntiles = p.tiles_m * p.tiles_n;
  int id = blockIdx.x;
  if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
  const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int z = blockIdx.z;
  const char* Ab = p.A;
  const char* Bb = p.B;
  char* Cb = p.C;
  int kbeg = 0, kend = p.K;
  long long coff = 0;
  if (p.split_k > 1) {
    kbeg = z * p.kchunk;
    kend = min(p.K, kbeg + p.kchunk);
    coff = (long long)z * p.slab_stride;
  } else {
    const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
    Ab += (zo * p.sA_o + zi * p.sA_i) * ET<T>::SZ;
    Bb += (zo * p.sB_o + zi * p.sB_i) * ET<T>::SZ;
    coff = zo * p.sC_o + zi * p.sC_i;
  }

  constexpr int ROLE_A = GATHER;  // 1/2: K-major A rows gathered; 3: MN-major A (k = output pixel) gathered
  constexpr int ROLE_B = 0;
  Loader<T, BM, AL, ROLE_A> la;
  Loader<T, BN, BL, ROLE_B> lb;
  la.init(p, Ab, p.lda, m0, p.M);
  lb.init(p, Bb, p.ldb, n0, p.N);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (kend - kbeg + BK - 1) / BK;
  if (nkt > 0) {
    la.issue(p, kbeg, kend);
    lb.issue(p, kbeg, kend);
    la.commit(smem);
    lb.commit(smem + GA::BYTES);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    char* cur = smem + (kt & 1) * STAGE;
    char* nxt = smem + ((kt + 1) & 1) * STAGE;
    const bool more = kt + 1 < nkt;
    if (more) {
      la.issue(p, kbeg + (kt + 1) * BK, kend);
      lb.issue(p, kbeg + (kt + 1) * BK, kend);
    }
    const char* sa = cur;
    const char* sb = cur + GA::BYTES;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_bf16<BM, AL>(sa, wm * TM + i, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_bf16<BN, BL>(sb, wn * TN + j, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_f32<BM, AL>(sa, wm * TM + i, u, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_f32<BN, BL>(sb, wn * TN + j, u, lane);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((&fa[i].x)[jj], (&fb[j].x)[jj], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      la.commit(nxt);
      lb.commit(nxt + GA::BYTES);
    }
    __syncthreads();
  }

  gemm_epilogue<T, TM, TN, 2, BN, NTHREADS>(acc, p, Cb, coff, m0 + wm * TM * 32, n0 + wn * TN * 32, wm, n0, tile_m, lane, smem);
}

// Deterministic split-K, second launch: C[m][n] (+)= slab[0][m][n] + slab[1][m][n] + ... in ascending K order.
__global__ __launch_bounds__(NTHREADS) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N,
                                                                 long long ldc, int nsl, int accumulate) {
  const long long mn = (long long)M * N;
  const int n4 = N >> 2;   // N % 4 == 0 (checked by the launcher)
  for (long long i = (long long)blockIdx.x * NTHREADS + threadIdx.x; i < (long long)M * n4; i += (long long)gridDim.x * NTHREADS) {
    const int m = (int)(i / n4), c = (int)(i - (long long)m * n4);
    const float4* src = reinterpret_cast<const float4*>(ws + (long long)m * N) + c;
    float4 a = *src;
    int z = 1;
    // eight slabs' loads in flight, added in ascending K order as before (one load per round trip held the 40-slab sums of the
    // layer-1 conv weight gradients at 0.8 TB/s)
    for (; z + 8 <= nsl; z += 8) {
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (z + u) * mn);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w;
      }
    }
    for (; z < nsl; ++z) {
      const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + z * mn);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float* dst = C + (long long)m * ldc + 4 * c;
    if (accumulate) {
      dst[0] += a.x; dst[1] += a.y; dst[2] += a.z; dst[3] += a.w;
    } else {
      dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    }
  }
}

template <typename T, int BM, int BN, int AL, int BL, int GATHER>
int launch(const KParams& p, int zdim, hipStream_t st) {
  constexpr int smem = 2 * (TileGeom<T, BM, AL>::BYTES + TileGeom<T, BN, BL>::BYTES);
  static bool attr_done = false;
  auto kern = gemm_kernel<T, BM, BN, AL, BL, GATHER>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
      return -2;
    }
    attr_done = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, 1, zdim);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), smem, st, p);
  set_last_kernel("gemm_kernel<%s, %d, %d, %d, %d, %d>", sizeof(T) == 4 ? "float" : "bf16_t", BM, BN, AL, BL, GATHER);
  return check_launch("gemm_kernel");
}

template <typename T, int BM, int BN>
int dispatch_layout(const HtrvtGemmDesc* d, const KParams& p, int zdim, hipStream_t st) {
  const int al = d->a_layout, bl = d->b_layout, g = d->gather;
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 0) return launch<T, BM, BN, 0, 0, 0>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 1) return launch<T, BM, BN, 0, 0, 1>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_KMAJOR && g == 2) return launch<T, BM, BN, 0, 0, 2>(p, zdim, st);
  if (al == HTRVT_KMAJOR && bl == HTRVT_MNMAJOR && g == 0) return launch<T, BM, BN, 0, 1, 0>(p, zdim, st);
  if (al == HTRVT_MNMAJOR && bl == HTRVT_MNMAJOR && g == 0) return launch<T, BM, BN, 1, 1, 0>(p, zdim, st);
  if (al == HTRVT_MNMAJOR && bl == HTRVT_MNMAJOR && g == 3) return launch<T, BM, BN, 1, 1, 3>(p, zdim, st);
  set_error("htrvt_gemm: unsupported layout/gather combination (a=%d b=%d gather=%d)", al, bl, g);
  return -1;
}

int pick_tile(const HtrvtGemmDesc* d, int* bm, int* bn) {
  *bm = 128;
  if (d->tile >= 1000) {
    *bm = d->tile / 1000;
    *bn = d->tile % 1000;
  } else if (d->N <= 64) {
    *bn = 64;
  } else if (d->N % 192 == 0 && d->N % 128 != 0) {
    *bn = 192;
  } else if (d->N % 192 == 0 && d->gather == HTRVT_GATHER_CONV_FWD) {
    *bn = 192;
  } else {
    *bn = 128;
  }
  if (*bm != 128 || (*bn != 64 && *bn != 128 && *bn != 192)) {
    set_error("htrvt_gemm: tile %dx%d is not built", *bm, *bn);
    return -1;
  }
  return 0;
}

}  // namespace

extern "C" int htrvt_gemm_num_mtiles(const HtrvtGemmDesc* d) {
  if (d->dtype == HTRVT_BF16 && d->tile != 1) {
    const int r = gemm_dma_num_mtiles(d);
    if (r > 0) return r;
  }
  int bm, bn;
  if (pick_tile(d, &bm, &bn)) return -1;
  return (d->M + bm - 1) / bm;
}

namespace htrvt {
int gemm_halo_s2_try_launch(const HtrvtGemmDesc*, KParams&, hipStream_t, bool probe);
}

// number of M tiles (= rows of a fused BatchNorm-backward partial buffer) of the merged strided-dgrad launch `d`
// (cls_h = cls_w = -2), or 0 when that form does not serve it and the caller launches one parity class at a time
extern "C" int htrvt_gemm_dgrad_merged_tiles(const HtrvtGemmDesc* d) {
  if (d == nullptr) return 0;
  KParams p;
  const int r = gemm_halo_s2_try_launch(d, p, nullptr, true);
  return r > 0 ? r : 0;
}

static int launch_main(const HtrvtGemmDesc* d, KParams& p, int bm, int bn, int zdim, hipStream_t st);

// every operand of the launch (A / the gathered tensor, B, C and its same-shaped side inputs) inside one 2 GiB buffer descriptor
static bool operands_below_2gib(const HtrvtGemmDesc* d) {
  const long long lim = (1ll << 31) - 64, es = d->dtype == HTRVT_BF16 ? 2 : 4;
  long long a, b, c;
  if (d->gather == HTRVT_GATHER_CONV_FWD || d->gather == HTRVT_GATHER_CONV_WGRAD) a = (long long)d->nB * d->Hi * d->Wi * d->Ci * es;
  else if (d->gather == HTRVT_GATHER_CONV_DGRAD) a = (long long)d->nB * d->Ho * d->Wo * d->Co * es;
  else a = (d->a_layout == HTRVT_KMAJOR ? (long long)d->M : (long long)d->K) * d->lda * es;
  if (d->gather == HTRVT_GATHER_CONV_WGRAD) b = (long long)d->K * d->ldb * es;
  else b = (d->b_layout == HTRVT_KMAJOR ? (long long)d->N : (long long)d->K) * d->ldb * es;
  const long long crows = (d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h >= 0) ? (long long)d->nB * d->Hi * d->Wi : d->M;
  c = crows * d->ldc * (d->c_f32 ? 4 : es);
  return a < lim && b < lim && c < lim;
}

extern "C" int htrvt_gemm(const HtrvtGemmDesc* d, void* stream) {
  HTRVT_REQUIRE(d != nullptr, "htrvt_gemm: null descriptor");
  HTRVT_REQUIRE(d->dtype == HTRVT_F32 || d->dtype == HTRVT_BF16, "htrvt_gemm: bad dtype %d", d->dtype);
  const bool cls = d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h >= 0;
  // cls_h == -2: every parity class of a strided dgrad in ONE launch (gemm_halo.hip: gemm_halo_s2_try_launch); M = all input
  // pixels, K = (taps + 1 with A2) * Cpad
  const bool merged = d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h == -2;
  HTRVT_REQUIRE(d->cls_h >= -2 && (d->cls_h != -2 || merged), "htrvt_gemm: cls_h = %d", d->cls_h);
  HTRVT_REQUIRE(d->M > 0 && d->N > 0 && (d->K > 0 || (cls && d->K == 0)), "htrvt_gemm: empty problem M=%d N=%d K=%d", d->M,
                d->N, d->K);
  HTRVT_REQUIRE(!cls || (d->dtype == HTRVT_BF16 && d->tile != 1 && d->cls_h < d->sh && d->cls_w >= 0 && d->cls_w < d->sw),
                "htrvt_gemm: parity-class dgrad needs bfloat16 and 0 <= cls < stride");
  HTRVT_REQUIRE(d->A && d->B && d->C, "htrvt_gemm: null operand");
  const int ch = d->dtype == HTRVT_BF16 ? 8 : 4, bk = d->dtype == HTRVT_BF16 ? 64 : 32;
  // every 16-byte chunk must be fully inside or fully outside an operand row
  if (d->gather == 0) {
    if (d->a_layout == HTRVT_KMAJOR)
      HTRVT_REQUIRE(d->K % ch == 0 && d->lda % ch == 0, "htrvt_gemm: K/lda must be multiples of %d", ch);
    else
      HTRVT_REQUIRE(d->M % ch == 0 && d->lda % ch == 0, "htrvt_gemm: M/lda must be multiples of %d (MN-major A)", ch);
    if (d->b_layout == HTRVT_KMAJOR)
      HTRVT_REQUIRE(d->K % ch == 0 && d->ldb % ch == 0, "htrvt_gemm: K/ldb must be multiples of %d", ch);
    else
      HTRVT_REQUIRE(d->N % ch == 0 && d->ldb % ch == 0, "htrvt_gemm: N/ldb must be multiples of %d (MN-major B)", ch);
  } else {
    HTRVT_REQUIRE(d->Cpad > 0 && d->Cpad % bk == 0, "htrvt_gemm: Cpad=%d must be a multiple of %d", d->Cpad, bk);
    HTRVT_REQUIRE((d->sh == 1 || d->sh == 2) && (d->sw == 1 || d->sw == 2), "htrvt_gemm: conv stride must be 1 or 2");
    HTRVT_REQUIRE(d->Ci % ch == 0 && d->Co % ch == 0, "htrvt_gemm: conv channels must be multiples of %d", ch);
    const long long taps = (long long)d->kh * d->kw;
    if (d->gather == HTRVT_GATHER_CONV_FWD)
      HTRVT_REQUIRE(d->M == d->nB * d->Ho * d->Wo && d->K == taps * d->Cpad && d->N == d->Co && d->Cpad >= d->Ci,
                    "htrvt_gemm: conv fwd extents inconsistent");
    if (d->gather == HTRVT_GATHER_CONV_DGRAD && !cls)
      HTRVT_REQUIRE(d->M == d->nB * d->Hi * d->Wi && d->K == (taps + ((merged && d->A2 != nullptr) ? 1 : 0)) * d->Cpad && d->N == d->Ci && d->Cpad >= d->Co,
                    "htrvt_gemm: conv dgrad extents inconsistent");
    if (d->gather == HTRVT_GATHER_CONV_WGRAD)
      HTRVT_REQUIRE(d->K == d->nB * d->Ho * d->Wo && d->M == taps * d->Cpad && d->N == d->Co && d->Cpad >= d->Ci &&
                        d->a_layout == HTRVT_MNMAJOR && d->b_layout == HTRVT_MNMAJOR,
                    "htrvt_gemm: conv wgrad extents inconsistent");
  }
  HTRVT_REQUIRE(!(d->split_k > 1) || (d->accumulate && d->c_f32 && d->batch <= 1),
                "htrvt_gemm: split_k needs accumulate=1, c_f32=1, batch<=1");
  HTRVT_REQUIRE(!d->accumulate || d->c_f32, "htrvt_gemm: accumulate needs c_f32");
  HTRVT_REQUIRE(d->splitk_ws == nullptr || (d->N % 4 == 0 && (reinterpret_cast<unsigned long long>(d->splitk_ws) & 15) == 0),
                "htrvt_gemm: splitk_ws needs N %% 4 == 0 and a 16-byte aligned workspace");

  int bm, bn;
  if (pick_tile(d, &bm, &bn)) return -1;
  KParams p;
  p.A = (const char*)d->A;
  p.B = (const char*)d->B;
  p.C = (char*)d->C;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc;
  p.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
  p.sA_o = d->sA_o; p.sA_i = d->sA_i; p.sB_o = d->sB_o; p.sB_i = d->sB_i; p.sC_o = d->sC_o; p.sC_i = d->sC_i;
  p.split_k = d->split_k > 1 ? d->split_k : 1;
  p.kchunk = d->K;
  if (p.split_k > 1) {
    int per = (d->K + p.split_k - 1) / p.split_k;
    p.kchunk = ((per + bk - 1) / bk) * bk;
    p.split_k = (d->K + p.kchunk - 1) / p.kchunk;
  }
  p.nB = d->nB; p.Hi = d->Hi; p.Wi = d->Wi; p.Ci = d->Ci; p.Ho = d->Ho; p.Wo = d->Wo; p.Co = d->Co;
  p.kh = d->kh; p.kw = d->kw; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw; p.Cpad = d->Cpad;
  p.alpha = d->alpha;
  p.act = d->act; p.c_f32 = d->c_f32 || d->dtype == HTRVT_F32; p.accumulate = d->accumulate;
  p.bias = d->bias; p.colscale = d->colscale; p.preact = (char*)d->preact; p.residual = (const char*)d->residual; p.colstats = d->colstats;
  p.tiles_m = (d->M + bm - 1) / bm;
  p.tiles_n = (d->N + bn - 1) / bn;
  p.wo_shift = p.howo_shift = p.wq_shift = p.hwq_shift = -1;
  p.relu_src = (const char*)d->relu_src;
  p.relu_bits = d->relu_bits;
  HTRVT_REQUIRE(d->relu_bits == 0 || (d->relu_src != nullptr && d->bnb_partial[0] != nullptr && d->relu_scale == nullptr &&
                                      d->gather == HTRVT_GATHER_CONV_DGRAD && d->dtype == HTRVT_BF16 && d->batch <= 1 && (d->ldc & 7) == 0 &&
                                      !(d->residual == nullptr && d->bnb_partial[1] != nullptr)),
                "htrvt_gemm: relu_bits needs a bfloat16 conv dgrad with relu_src, one BatchNorm sum set (or a residual and two), ldc a multiple of 8");
  for (int t = 0; t < 2; ++t) {
    p.bnb_x[t] = (const char*)d->bnb_x[t];
    p.bnb_mean[t] = d->bnb_mean[t];
    p.bnb_rstd[t] = d->bnb_rstd[t];
    p.bnb_partial[t] = d->bnb_partial[t];
  }
  p.bnb_tile0 = d->bnb_tile0;
  p.relu_sc = d->relu_scale;
  p.relu_sf = d->relu_shift;
  HTRVT_REQUIRE((d->relu_scale == nullptr) == (d->relu_shift == nullptr), "htrvt_gemm: relu_scale and relu_shift go together");
  HTRVT_REQUIRE(d->relu_scale == nullptr || (d->bnb_partial[0] != nullptr && d->bnb_partial[1] == nullptr && d->residual == nullptr &&
                                             d->relu_src == nullptr && d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h < 0),
                "htrvt_gemm: relu_scale / relu_shift need exactly one bnb set, no residual, no relu_src, an unstrided conv dgrad");
  const bool fused_bwd = d->relu_src != nullptr || d->bnb_partial[0] != nullptr;
  p.cls_h = p.cls_w = -1;
  p.extra_off = 0;
  HTRVT_REQUIRE(d->A2 == nullptr || cls || merged, "htrvt_gemm: A2 needs a parity-class dgrad launch");
  p.Hq = d->Hi; p.Wq = d->Wi;
  p.ntapsel = d->kh * d->kw;
  for (int t = 0; t < 12; ++t) p.tapsel[t] = (unsigned char)t;
  if (cls) {  // taps (dy,dx) that reach pixels of this class: (cls + pad - d) divisible by the stride
    p.cls_h = d->cls_h; p.cls_w = d->cls_w;
    p.Hq = (d->Hi - d->cls_h + d->sh - 1) / d->sh;
    p.Wq = (d->Wi - d->cls_w + d->sw - 1) / d->sw;
    p.ntapsel = 0;
    for (int dy = 0; dy < d->kh; ++dy)
      for (int dx = 0; dx < d->kw; ++dx)
        if ((d->cls_h + d->ph - dy) % d->sh == 0 && (d->cls_w + d->pw - dx) % d->sw == 0)
          p.tapsel[p.ntapsel++] = (unsigned char)(dy * d->kw + dx);
    if (d->A2 != nullptr) {
      HTRVT_REQUIRE(d->cls_h == 0 && d->cls_w == 0, "htrvt_gemm: A2 (1x1 downsample gradient) goes with the class (0, 0) launch only");
      const long long diff = (const char*)d->A2 - (const char*)d->A;
      HTRVT_REQUIRE(diff > 0 && diff + (long long)d->nB * d->Ho * d->Wo * d->Co * 2 < (1ll << 31) - 64,
                    "htrvt_gemm: A2 must lie behind A, both inside 2 GiB");
      p.tapsel[p.ntapsel++] = (unsigned char)(d->kh * d->kw);       // one more tap: rows kh*kw of the packed weight
      p.extra_off = (unsigned)diff;
    }
    HTRVT_REQUIRE(d->M == d->nB * p.Hq * p.Wq && d->K == p.ntapsel * d->Cpad && d->N == d->Ci,
                  "htrvt_gemm: parity-class dgrad extents inconsistent (expected M=%d K=%d)", d->nB * p.Hq * p.Wq,
                  p.ntapsel * d->Cpad);
  }
  p.tappack = 0;
  for (int t = 0; t < 12; ++t) p.tappack |= (unsigned long long)(p.tapsel[t] & 15) << (4 * t);
  const int zdim = p.split_k > 1 ? p.split_k : (d->batch > 1 ? d->batch : 1);
  HTRVT_REQUIRE((long long)p.tiles_m * p.tiles_n < (1ll << 31) && zdim < 65536, "htrvt_gemm: grid too large");
  hipStream_t st = (hipStream_t)stream;
  p.slab_stride = 0;
  if (p.split_k > 1 && d->splitk_ws != nullptr) {
    // reproducible form: every K range owns a dense [M][N] slab, summed in K order by splitk_reduce_kernel below
    p.C = reinterpret_cast<char*>(d->splitk_ws);
    p.ldc = d->N;
    p.accumulate = 0;
    p.slab_stride = (long long)d->M * d->N;
    const int rc = launch_main(d, p, bm, bn, zdim, st);
    if (rc) return rc;
    long long items = (long long)d->M * (d->N / 4);
    int grid = (int)((items + NTHREADS - 1) / NTHREADS);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(NTHREADS), 0, st, d->splitk_ws, reinterpret_cast<float*>(d->C), d->M,
                       d->N, (long long)d->ldc, p.split_k, d->accumulate);
    return check_launch("splitk_reduce");
  }
  return launch_main(d, p, bm, bn, zdim, st);
}

// picks the kernel family (LDS-DMA bfloat16 tiles or the register-staged kernel) and launches it
static int launch_main(const HtrvtGemmDesc* d, KParams& p, int bm, int bn, int zdim, hipStream_t st) {
  const bool cls = d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h >= 0;
  const bool fused_bwd = d->relu_src != nullptr || d->bnb_partial[0] != nullptr;
  if (d->dtype == HTRVT_BF16 && d->tile != 1) {  // throughput path: 256-row tiles, operands by LDS-DMA
    KParams q = p;
    int r = gemm8p_try_launch(d, q, zdim, st);   // 8-phase schedule, epilogue from the accumulators (gemm8p.hip)
    if (r != 0) return r < 0 ? r : 0;
    q = p;
    r = gemm8pt_try_launch(d, q, zdim, st);      // the same schedule for MN-major operands: Linear weight gradients
    if (r != 0) return r < 0 ? r : 0;
    q = p;
    r = gemm_dma_try_launch(d, q, zdim, st);     // one barrier per k-tile, LDS-staged epilogue (gemm_dma.hip)
    if (r != 0) return r < 0 ? r : 0;
  }
  // Past this point the register-staged kernel (64-bit addressing, 128-row tiles) takes the launch.  Forms that exist in
  // the LDS-DMA families only -- whose operands go through 2 GiB buffer descriptors -- are REFUSED here, never rerouted:
  // a parity class / merged strided dgrad, fused backward epilogues, per-tile column sums sized for 256-row tiles, and any
  // explicit family selector (tile 2 .. 17) the caller asked for.  (Round 4's memory fault was such a reroute: the tripled
  // layer-1 operand of the split-bf16 path passed 2 GiB at 128 images, the LDS-DMA family declined, and the 128-row
  // fallback wrote 8 192 rows of column sums into a buffer htrvt_gemm_num_mtiles had sized for 4 096 -- 6 MB past its end.)
  HTRVT_REQUIRE(!(d->dtype == HTRVT_BF16 && d->tile >= 2 && d->tile <= 17 && !operands_below_2gib(d)),
                "htrvt_gemm: tile selector %d names an LDS-DMA kernel family; its operands must stay below 2 GiB: split the launch", d->tile);
  HTRVT_REQUIRE(!cls, "htrvt_gemm: parity-class dgrad is served by the LDS-DMA kernel only (M > 128, operands < 2 GiB)");
  HTRVT_REQUIRE(d->cls_h != -2, "htrvt_gemm: the merged strided dgrad (cls_h = -2) is served by the halo kernels only; ask htrvt_gemm_dgrad_merged_tiles first");
  // per-tile column sums: the caller sized `colstats` with htrvt_gemm_num_mtiles, i.e. for the 256-row tiles of the LDS-DMA
  // family; when that family declines the launch (an operand of 2 GiB or more) the 128-row tiles below would write twice as
  // many rows -- refuse instead of overrunning the buffer (the engine splits such a launch along the batch)
  HTRVT_REQUIRE(!(d->colstats != nullptr && d->dtype == HTRVT_BF16 && d->tile != 1 && gemm_dma_num_mtiles(d) > 0),
                "htrvt_gemm: column sums of a bfloat16 launch need the LDS-DMA kernels (operands below 2 GiB): split the batch");
  HTRVT_REQUIRE(!fused_bwd, "htrvt_gemm: relu_src / bnb_* need the bfloat16 LDS-DMA kernel with loader waves");
  if (d->dtype == HTRVT_BF16) {
    if (bn == 64) return dispatch_layout<bf16_t, 128, 64>(d, p, zdim, st);
    if (bn == 128) return dispatch_layout<bf16_t, 128, 128>(d, p, zdim, st);
    return dispatch_layout<bf16_t, 128, 192>(d, p, zdim, st);
  }
  if (bn == 64) return dispatch_layout<float, 128, 64>(d, p, zdim, st);
  if (bn == 128) return dispatch_layout<float, 128, 128>(d, p, zdim, st);
  return dispatch_layout<float, 128, 192>(d, p, zdim, st);
}
