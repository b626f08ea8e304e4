// gemm8p.hip -- host-side selection of the 8-phase bfloat16 GEMM kernels (gemm8p_impl.h; instantiated in
// gemm8p_plain.hip / gemm8p_conv.hip).  Called by htrvt_gemm (gemm.hip) before the older LDS-DMA kernels.
#include <stdlib.h>

#include "gemm_common.h"

using namespace htrvt;

namespace htrvt {
int gemm8p_dispatch_plain(int bn, int epi, const KParams&, int, hipStream_t);
int gemm8pp_dispatch_plain(int bn, int epi, const KParams&, int nwg, hipStream_t);
int gemm8pt_dispatch(const KParams&, int zdim, hipStream_t);
int gemm8p_dispatch_conv(int bn, int gather, int epi, const KParams&, int, hipStream_t);
}  // namespace htrvt

namespace {

constexpr int E_RES = 1, E_GELU = 2, E_GELUGRAD = 4, E_CSTATS = 8, E_RELUMASK = 16, E_BNB1 = 32, E_BNB2 = 64, E_F32 = 128,
              E_SCALE_RELU = 256;

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

bool extents_ok(const HtrvtGemmDesc* d) {   // every byte offset the loaders form must stay below 2^31
  const long long lim = (1ll << 31) - 64;
  long long a, b = (long long)d->N * d->ldb * 2;
  if (d->gather == HTRVT_GATHER_CONV_FWD) a = (long long)d->nB * d->Hi * d->Wi * d->Ci * 2;
  else if (d->gather == HTRVT_GATHER_CONV_DGRAD) a = (long long)d->nB * d->Ho * d->Wo * d->Co * 2;
  else a = (long long)d->M * d->lda * 2;
  // C and its same-shaped side inputs are addressed through 2 GiB buffer descriptors as well
  const long long crows = (d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h >= 0) ? (long long)d->nB * d->Hi * d->Wi : d->M;
  const long long c = crows * d->ldc * 2;
  return a < lim && b < lim && c < lim;
}

}  // namespace

namespace htrvt {

// tile selector (HtrvtGemmDesc.tile): 0 auto, 9 this family (auto width), 10 / 11 this family with 256 / 192 columns;
// 1..8 and BM*1000+BN keep meaning the older kernels
int gemm8p_pick_bn(const HtrvtGemmDesc* d) {
  if (d->tile == 10 || d->tile == 14) return 256;     // 13 / 14 / 15: the persistent form (auto width / 256 / 192 columns)
  if (d->tile == 11 || d->tile == 15) return 192;
  // rounds of 256 workgroups x columns per tile; the 192-column tile costs ~12 % more per FLOP (12 instead of 16 MFMAs per
  // phase, 14 % more operand bytes per FLOP).  Measured (tools/bench_gemm.py --only enc --tiles 4 10 11, M = 32768):
  // N = 768 -> 192 (512 tiles = 2 full rounds; 256 columns: 384 tiles = 1.5), N = 2304 / 3072 -> 256
  const long long tm = (d->M + 255) / 256;
  auto cost = [&](int bn, double per_flop) {
    const long long tiles = tm * ((d->N + bn - 1) / bn) * (d->batch > 1 ? d->batch : 1);
    return (double)((tiles + 255) / 256) * bn * per_flop;
  };
  return cost(192, 1.12) < cost(256, 1.0) ? 192 : 256;
}

bool gemm8p_serves(const HtrvtGemmDesc* d) {
  // auto (tile 0): the Linear shapes only.  The implicit-GEMM convolutions stay on the loader-wave kernels of
  // gemm_dma_impl.h unless this family is asked for: with all eight waves forming gather addresses the k-tile is
  // 15-25 % slower there (tools/bench_gemm.py --only conv --tiles 4 10 11: layer-1 forward 969 vs 813 TFLOP/s)
  if (!((d->tile == 0 && d->gather == HTRVT_GATHER_NONE) || (d->tile >= 9 && d->tile <= 11) || (d->tile >= 13 && d->tile <= 15 && d->gather == HTRVT_GATHER_NONE))) return false;
  if (d->dtype != HTRVT_BF16 || d->M <= 128) return false;
  if (d->cls_h == -2) return false;                       // merged strided dgrad: halo kernels only (gemm_halo.hip)
  if (gemm_small_m_prefers_bn128(d)) return false;      // few rows, narrow N: more, narrower tiles (gemm_dma.hip pick_bn)
  if (d->a_layout != HTRVT_KMAJOR || d->b_layout != HTRVT_KMAJOR) return false;
  if (d->gather != HTRVT_GATHER_NONE && d->gather != HTRVT_GATHER_CONV_FWD && d->gather != HTRVT_GATHER_CONV_DGRAD) return false;
  if (d->split_k > 1 || d->accumulate || d->A2 != nullptr) return false;
  // float32 C: plain Linear products only (bias at most): what the split-bf16 parity path asks for (engine.linear_fwd / _dgrad)
  if (d->c_f32 && (d->gather != HTRVT_GATHER_NONE || d->act != 0 || d->preact != nullptr || d->residual != nullptr || d->colstats != nullptr ||
                   d->colscale != nullptr || d->relu_src != nullptr || d->bnb_partial[0] != nullptr || d->batch > 1 ||
                   (long long)d->M * d->ldc * 4 >= (1ll << 31) - 64))
    return false;
  if (!extents_ok(d)) return false;
  const int bn = gemm8p_pick_bn(d);
  const int cw = bn == 256 ? 8 : 12;     // consecutive columns a lane stores
  if (d->N % cw) return false;
  // 16-byte row pieces: rows and batch slices of C (and of every same-shaped side input) must start on 8-byte boundaries
  // at least (the 24-byte pieces of the 192-column tile are dword-aligned dwordx4 + dwordx2 accesses)
  if ((d->ldc & 3) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return false;
  if (d->batch > 1 && ((d->sC_o | d->sC_i) & 3)) return false;
  if (d->act == 2 && d->preact == nullptr) return false;
  if ((d->colscale != nullptr || d->act == 3) && d->gather != HTRVT_GATHER_CONV_FWD) return false;
  if ((d->relu_src != nullptr || d->bnb_partial[0] != nullptr) && d->gather != HTRVT_GATHER_CONV_DGRAD) return false;
  if (d->relu_scale != nullptr || d->relu_bits != 0) return false;       // mask recomputed from the BatchNorm input / bit mask: staged epilogue only
  if (d->colstats != nullptr && d->gather != HTRVT_GATHER_CONV_FWD) return false;
  if (d->act != 1 && d->act != 2 && d->preact != nullptr) return false;   // pre-activation without GELU: not built
  return true;
}

// MN-major x MN-major plain products with float32 output -- the Linear weight gradients dW = dy^T x -- on the 8-phase
// schedule with transposed fragment reads (gemm8pt_impl.h).  tile 0 (auto) and 16; 3 / 4 / 6 keep naming the older kernels.
bool gemm8pt_serves(const HtrvtGemmDesc* d) {
  if (d->tile != 0 && d->tile != 16) return false;
  static const bool off = getenv("HTRVT_NO_MNMAJOR_8PHASE") != nullptr && getenv("HTRVT_NO_MNMAJOR_8PHASE")[0] == '1';   // A/B runs on one box
  if (off && d->tile == 0) return false;
  if (d->dtype != HTRVT_BF16 || d->gather != HTRVT_GATHER_NONE || d->a_layout != HTRVT_MNMAJOR || d->b_layout != HTRVT_MNMAJOR) return false;
  if (!d->c_f32 || d->batch > 1 || d->M < 256 || d->N < 256 || d->K < 256) return false;
  // auto: outputs of at least 16 tiles of 256 x 256.  The proj weight gradient (768 x 768: 9 tiles) needs a 20-28-way K split to
  // fill the chip, i.e. 18-25 k-tiles per workgroup and a 28-slab sum: measured 626 TFLOP/s at best against 741 on the
  // one-barrier kernel's 256 x 192 tiles at split 16 (tools/bench_gemm.py --only lwgrad --tiles 0 3)
  if (d->tile == 0 && (long long)((d->M + 255) / 256) * ((d->N + 255) / 256) < 16) return false;
  if ((d->M & 7) || (d->N & 7) || (d->lda & 7) || (d->ldb & 7) || (d->ldc & 3)) return false;
  if ((reinterpret_cast<unsigned long long>(d->A) & 15) || (reinterpret_cast<unsigned long long>(d->B) & 15) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return false;
  if (d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->residual != nullptr || d->colstats != nullptr || d->colscale != nullptr) return false;
  const long long lim = (1ll << 31) - 64;
  if ((long long)d->K * d->lda * 2 >= lim || (long long)d->K * d->ldb * 2 >= lim) return false;
  return true;
}

int gemm8pt_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st) {
  if (!gemm8pt_serves(d)) return 0;
  // plain stores only: the whole K in one launch without accumulation, or K ranges into slabs (summed by splitk_reduce_kernel)
  if (p.split_k > 1 ? p.slab_stride == 0 : p.accumulate != 0) return 0;
  p.tiles_m = (d->M + 255) / 256;
  p.tiles_n = (d->N + 255) / 256;
  return gemm8pt_dispatch(p, zdim, st);
}

// returns 1 if it launched, 0 if this family has no kernel for the call, < 0 on error
int gemm8p_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st) {
  if (!gemm8p_serves(d)) return 0;
  const int bn = gemm8p_pick_bn(d);
  int epi = 0;
  if (d->residual != nullptr) epi |= E_RES;
  if (d->act == 1) epi |= E_GELU;
  if (d->act == 2) epi |= E_GELUGRAD;
  if (d->colstats != nullptr) epi |= E_CSTATS;
  if (d->relu_src != nullptr) epi |= E_RELUMASK;
  if (d->bnb_partial[0] != nullptr) epi |= E_BNB1;
  if (d->bnb_partial[1] != nullptr) epi |= E_BNB2;
  if (d->colscale != nullptr || d->act == 3) epi |= E_SCALE_RELU;
  if (d->c_f32) epi |= E_F32;
  p.tiles_m = (d->M + 255) / 256;
  p.tiles_n = (d->N + bn - 1) / bn;
  p.wo_shift = p.howo_shift = -1;
  p.wq_shift = p.hwq_shift = -1;
  if (d->cls_h >= 0) {
    const int a = ilog2_exact(p.Wq), b = ilog2_exact(p.Hq * p.Wq);
    if (a >= 0 && b >= 0) {
      p.wq_shift = a;
      p.hwq_shift = b;
    }
  }
  if (d->gather == HTRVT_GATHER_NONE) {
    // Persistent walk (gemm8pp_impl.h): one workgroup per CU, the DMA stream and the k loop run through the tile boundaries,
    // the epilogue of a tile is folded into the first k-tile of the next.  tile 0 (auto) and 13; 9-11 keep naming the
    // one-tile-per-workgroup kernels (A/B runs, tests).  HTRVT_NO_PERSISTENT_GEMM=1 switches the auto route off.
    static const bool off = getenv("HTRVT_NO_PERSISTENT_GEMM") != nullptr && getenv("HTRVT_NO_PERSISTENT_GEMM")[0] == '1';
    const long long ntiles = (long long)p.tiles_m * p.tiles_n;
    // auto: the plain / bias epilogue only.  With the GELU epilogue the folded flush is VALU-bound -- one wave per SIMD
    // evaluates erf while its partner's 16 MFMAs are long done -- and measured 2-3 % SLOWER than the unfolded epilogue, in
    // which both waves of a SIMD share the VALU (tools/bench_gemm.py --only enc --tiles 9 0: fc1 forward 830 vs 808 TFLOP/s)
    // round 5: bias + residual too (192-column tiles: proj / fc2 forward), the residual through registers (gemm8pp_impl.h)
    // (* GELU'(saved pre-activation) through the same register loads was built and measured 7 % SLOWER than the one-tile kernel:
    // its flush is VALU-bound on one wave per SIMD, profiles/r05_experiments.md (g))
    static const bool res_off = getenv("HTRVT_NO_PERSISTENT_RES") != nullptr && getenv("HTRVT_NO_PERSISTENT_RES")[0] == '1';
    const bool res_ok = epi == E_RES && bn == 192 && !res_off && d->K >= 256 && (reinterpret_cast<unsigned long long>(d->residual) & 3) == 0;
    const bool want = (d->tile >= 13 && d->tile <= 15) || (d->tile == 0 && !off && (epi == 0 || res_ok));
    if (want && (epi == 0 || epi == E_GELU || res_ok) && zdim == 1 && d->batch <= 1 && d->K % 128 == 0 && d->K >= 256 &&
        (reinterpret_cast<unsigned long long>(d->bias) & 15) == 0 && (d->preact == nullptr || (reinterpret_cast<unsigned long long>(d->preact) & 15) == 0)) {
      static int ncu = 0;
      if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount & ~7;
        if (ncu < 8) ncu = 256;
      }
      if (ntiles >= ncu) {
        const int r = gemm8pp_dispatch_plain(bn, epi, p, ncu, st);
        if (r != 0) return r;
      }
    }
    return gemm8p_dispatch_plain(bn, epi, p, zdim, st);
  }
  return gemm8p_dispatch_conv(bn, d->gather, epi, p, zdim, st);
}

}  // namespace htrvt
