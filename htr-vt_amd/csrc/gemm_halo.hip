// gemm_halo.hip -- instantiations + host-side eligibility of the halo-staged 3x3 stride-1 convolution kernels
// (gemm_halo_impl.h); called from gemm_dma_try_launch before the generic gather kernels.
#include "gemm_halo_impl.h"

namespace htrvt {

// 1 launched, 0 not served, < 0 error.  p.tiles_m / tiles_n / shifts are set by the caller for a 256 x bn tiling.
int gemm_halo_try_launch(const HtrvtGemmDesc* d, const KParams& p, int bn, hipStream_t st) {
  const bool fwd = d->gather == HTRVT_GATHER_CONV_FWD, dgr = d->gather == HTRVT_GATHER_CONV_DGRAD;
  if (!(fwd || dgr) || d->dtype != HTRVT_BF16) return 0;
  if (d->tile != 0 && d->tile != 4 && d->tile != 12) return 0;      // 12: this kernel where eligible; 5: the generic gather (A/B)
  if (d->kh != 3 || d->kw != 3 || d->sw != 1 || d->ph != 1 || d->pw != 1 || d->cls_h >= 0) return 0;
  if (d->sh != 1 && !(fwd && d->sh == 2)) return 0;                           // forward also with a row stride of 2 (layer1.0.conv1)
  if (d->Ho != (d->Hi - 1) / d->sh + 1 || d->Wo != d->Wi || (d->Wi % 256) != 0) return 0;    // an M tile = 256 pixels of one image row
  if (d->K != 9 * d->Cpad || d->batch > 1 || d->split_k > 1 || d->c_f32) return 0;
  if ((d->ldc & 7) || (d->N & 7) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return 0;   // the staged bf16 epilogue
  // forward: raw output (+ BatchNorm column sums) in training, or the eval-mode fold C = relu?(acc * colscale + bias [+ residual])
  // -- both are paths of the shared staged epilogue; GELU / saved pre-activations do not occur on convolutions
  if (d->preact != nullptr || (d->act != 0 && d->act != 3)) return 0;
  if (dgr && (d->colscale != nullptr || d->bias != nullptr || d->act != 0)) return 0;
  if (fwd && d->residual != nullptr && d->colscale == nullptr) return 0;   // a residual only as part of the eval fold
  if (bn == 192) return fwd ? launch_halo<192, false>(p, st) : launch_halo<192, true>(p, st);
  if (bn == 128) return fwd ? launch_halo<128, false>(p, st) : launch_halo<128, true>(p, st);
  return 0;
}

}  // namespace htrvt
