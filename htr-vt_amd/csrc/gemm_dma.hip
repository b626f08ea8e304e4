// gemm_dma.hip -- host-side selection of the bfloat16 LDS-DMA GEMM kernel variant (kernels: gemm_dma_impl.h,
// instantiated per N-tile width in gemm_dma_bn*.hip).
#include <stdlib.h>

#include "gemm_common.h"

using namespace htrvt;

namespace htrvt {
int gemm_dma_dispatch_bn64(const HtrvtGemmDesc*, const KParams&, int, hipStream_t, bool);
int gemm_dma_dispatch_bn128(const HtrvtGemmDesc*, const KParams&, int, hipStream_t, bool);
int gemm_dma_dispatch_bn128_s3(const HtrvtGemmDesc*, const KParams&, int, hipStream_t, bool);
int gemm_dma_dispatch_bn192(const HtrvtGemmDesc*, const KParams&, int, hipStream_t, bool);
int gemm_dma_dispatch_bn256(const HtrvtGemmDesc*, const KParams&, int, hipStream_t, bool);
int gemm_halo_try_launch(const HtrvtGemmDesc*, const KParams&, int bn, hipStream_t);
int gemm_halo_fs2_try_launch(const HtrvtGemmDesc*, const KParams&, int bn, hipStream_t);
int gemm_hwgrad_try_launch(const HtrvtGemmDesc*, KParams&, int zdim, hipStream_t);
int gemm_halo_s2_try_launch(const HtrvtGemmDesc*, KParams&, hipStream_t, bool probe);
int conv1x1_try_launch(const HtrvtGemmDesc*, KParams&, hipStream_t);
}  // namespace htrvt

namespace {

constexpr int BM_ = 256;

// tile selector: 0 auto, 3: every wave loads, 4: 4 dedicated loader waves.  Measured (tools/bench_gemm.py --tiles 3 4):
// loader waves win 5-15 % on the conv forward/dgrad gathers and on K >= 2048 plain GEMMs, lose on MN-major A.
bool use_loader_waves(const HtrvtGemmDesc* d) {
  if (d->tile == 3) return false;
  if (d->tile == 4 || d->tile == 5 || d->tile == 12) return true;
  if (d->a_layout != HTRVT_KMAJOR) return false;
  return d->gather == HTRVT_GATHER_CONV_FWD || d->gather == HTRVT_GATHER_CONV_DGRAD || d->K >= 2048;
}

int pick_bn(const HtrvtGemmDesc* d) {
  const int N = d->N;
  const bool fused = d->relu_src != nullptr || d->bnb_partial[0] != nullptr;
  // 256x256 tiles move 14 % fewer operand bytes per FLOP through the (per-CU, ~70 GB/s) LDS-DMA path than 256x192
  // (256x384 would save 29 % but its 192 accumulator registers per wave do not fit beside the loader state)
  // measured (tools/bench_gemm.py --tiles 3 4 6): +8 % on the N=768 conv wgrad, +20 % on 4096^3, but -20 % where the
  // tile count stops filling whole rounds of 256 CUs (N=768 / 2304 Linear layers) -> conv wgrad and explicit only
  if (N % 256 == 0 && !fused && d->colscale == nullptr && d->act != 3 && (d->tile == 6 || (d->tile == 0 && d->gather == HTRVT_GATHER_CONV_WGRAD))) return 256;
  if (N <= 64) return 64;
  if (N <= 128 || d->tile == 7 || d->tile == 8 || gemm_small_m_prefers_bn128(d)) return 128;   // 7 / 8: experiment selectors, 128-column tiles with 2 / 3 stages
  const int p192 = (N + 191) / 192 * 192, p128 = (N + 127) / 128 * 128;
  // few output pixels (layer 3 at 16 images per GPU: 32 M tiles): 128-column tiles fill 192 instead of 128 CUs
  if ((d->gather == HTRVT_GATHER_CONV_FWD || d->gather == HTRVT_GATHER_CONV_DGRAD) && d->tile == 0 && d->cls_h < 0 && N >= 256 &&
      N % 128 == 0 && (long long)((d->M + BM_ - 1) / BM_) * (p192 / 192) <= 128)
    return 128;
  return p192 <= p128 ? 192 : 128;
}

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

// every byte offset the loaders form must stay below 2^31
bool extents_ok(const HtrvtGemmDesc* d) {
  const long long lim = (1ll << 31) - 64;
  auto plain = [&](int layout, long long rows, long long ld) {
    return layout == HTRVT_KMAJOR ? rows * ld * 2 : (long long)d->K * ld * 2;
  };
  long long a, b = plain(d->b_layout, d->N, d->ldb);
  if (d->gather == HTRVT_GATHER_CONV_FWD || d->gather == HTRVT_GATHER_CONV_WGRAD)
    a = (long long)d->nB * d->Hi * d->Wi * d->Ci * 2;
  else if (d->gather == HTRVT_GATHER_CONV_DGRAD)
    a = (long long)d->nB * d->Ho * d->Wo * d->Co * 2;
  else
    a = plain(d->a_layout, d->M, d->lda);
  return a < lim && b < lim;
}

}  // namespace

namespace htrvt {

int gemm_dma_num_mtiles(const HtrvtGemmDesc* d) {
  if (d->dtype != HTRVT_BF16 || d->M <= 128 || d->tile == 1) return -1;
  return (d->M + BM_ - 1) / BM_;
}

int gemm_dma_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st) {
  if (d->dtype != HTRVT_BF16 || d->M <= 128 || d->tile == 1 || !extents_ok(d)) return 0;
  if (d->gather == HTRVT_GATHER_CONV_DGRAD && d->cls_h == -2) return gemm_halo_s2_try_launch(d, p, st, false);   // all parity classes, one launch
  if (d->gather == HTRVT_GATHER_CONV_FWD && d->kh == 1 && d->kw == 1) {   // strided 1x1 downsample convolutions: HBM-rate streaming kernel (conv1x1.hip)
    static const bool off = getenv("HTRVT_NO_CONV1X1") != nullptr && getenv("HTRVT_NO_CONV1X1")[0] == '1';   // A/B runs on one box
    const int r = off ? 0 : conv1x1_try_launch(d, p, st);
    if (r != 0) return r;
  }
  if (d->relu_src != nullptr || d->bnb_partial[0] != nullptr) {
    // served only by the staged epilogue with a fixed column group per wave (12 waves, 6/4/2 column groups)
    if (!use_loader_waves(d) || d->c_f32 || (d->ldc & 7) || (d->N & 7)) return 0;
  }
  if (d->colscale != nullptr || d->act == 3) {
    // per-column scale / trailing ReLU exist in the staged bf16 epilogue of the conv-forward kernels only
    if (d->gather != HTRVT_GATHER_CONV_FWD || d->c_f32 || (d->ldc & 7) || (d->N & 7) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return 0;
    if (d->batch > 1 && ((d->sC_o | d->sC_i) & 7)) return 0;
  }
  if (d->gather == HTRVT_GATHER_CONV_WGRAD) {   // 3x3 stride-1 convolutions: halo-staged x operand (gemm_hwgrad_impl.h)
    const int r = gemm_hwgrad_try_launch(d, p, zdim, st);
    if (r != 0) return r;
  }
  const int bn = pick_bn(d);
  p.tiles_m = (d->M + BM_ - 1) / BM_;
  p.tiles_n = (d->N + bn - 1) / bn;
  p.wo_shift = p.howo_shift = -1;
  if (d->gather == HTRVT_GATHER_CONV_WGRAD) {
    const int a = ilog2_exact(d->Wo), b = ilog2_exact(d->Ho * d->Wo);
    if (a >= 0 && b >= 0) {
      p.wo_shift = a;
      p.howo_shift = b;
    }
  }
  p.wq_shift = p.hwq_shift = -1;
  if (d->cls_h >= 0) {
    const int a = ilog2_exact(p.Wq), b = ilog2_exact(p.Hq * p.Wq);
    if (a >= 0 && b >= 0) {
      p.wq_shift = a;
      p.hwq_shift = b;
    }
  }
  if (bn == 192 || bn == 128) {   // 3x3 stride-1 convolutions whose M tiles are row segments: halo-staged A operand (gemm_halo_impl.h)
    const int r = gemm_halo_try_launch(d, p, bn, st);
    if (r != 0) return r;
    const int r2 = gemm_halo_fs2_try_launch(d, p, bn, st);   // column stride 2: odd / even pixel images
    if (r2 != 0) return r2;
  }
  const bool spec = use_loader_waves(d);
  if (bn == 256) return gemm_dma_dispatch_bn256(d, p, zdim, st, false);
  if (bn == 64) return gemm_dma_dispatch_bn64(d, p, zdim, st, spec);
  if (bn == 128) return (d->tile == 8 || gemm_small_m_prefers_bn128(d)) ? gemm_dma_dispatch_bn128_s3(d, p, zdim, st, spec) : gemm_dma_dispatch_bn128(d, p, zdim, st, spec);
  return gemm_dma_dispatch_bn192(d, p, zdim, st, spec);
}

}  // namespace htrvt

