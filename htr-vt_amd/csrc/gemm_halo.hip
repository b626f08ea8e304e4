// gemm_halo.hip -- instantiations + host-side eligibility of the halo-staged 3x3 stride-1 convolution kernels
// (gemm_halo_impl.h); called from gemm_dma_try_launch before the generic gather kernels.
#include <stdlib.h>

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
  if (d->K != 9 * d->Cpad || d->batch > 1 || d->split_k > 1) return 0;
  if (d->c_f32) {
    // float32 C (the split-bf16 parity path's convolutions): plain stores from the accumulators, alpha, an optional float32 residual
    // on dgrad, per-tile column sums on the forward -- nothing else
    if (d->colscale != nullptr || d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->relu_src != nullptr ||
        d->bnb_partial[0] != nullptr || d->relu_scale != nullptr || d->accumulate || (fwd && d->residual != nullptr))
      return 0;
    if (bn == 192) return fwd ? launch_halo<192, false, true>(p, st) : launch_halo<192, true, true>(p, st);
    if (bn == 128) return fwd ? launch_halo<128, false, true>(p, st) : launch_halo<128, true, true>(p, st);
    return 0;
  }
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

// Forward with a column stride of 2 (layer2.0 / layer3.0 conv1, stride (2,2)): odd / even pixel images (gemm_halo_fs2_kernel).
// Same return convention and caller-set tiling as gemm_halo_try_launch.
int gemm_halo_fs2_try_launch(const HtrvtGemmDesc* d, const KParams& p, int bn, hipStream_t st) {
  if (d->gather != HTRVT_GATHER_CONV_FWD || d->dtype != HTRVT_BF16) return 0;
  if (d->tile != 0 && d->tile != 4 && d->tile != 12) return 0;      // 5: the generic gather (A/B)
  if (d->kh != 3 || d->kw != 3 || d->sw != 2 || (d->sh != 1 && d->sh != 2) || d->ph != 1 || d->pw != 1 || d->cls_h >= 0) return 0;
  if ((d->Wi & 1) || d->Wo != d->Wi / 2 || d->Ho != (d->Hi - 1) / d->sh + 1 || (d->Wo % 256) != 0) return 0;   // an M tile = 256 pixels of one output row
  if (d->K != 9 * d->Cpad || d->batch > 1 || d->split_k > 1 || d->M != d->nB * d->Ho * d->Wo) return 0;
  if (d->c_f32) {      // float32 C + per-tile column sums (the split-bf16 parity path), as in gemm_halo_try_launch
    if (d->colscale != nullptr || d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->residual != nullptr || d->accumulate) return 0;
    if (bn == 192) return launch_halo_fs2<192, true>(p, st);
    if (bn == 128) return launch_halo_fs2<128, true>(p, st);
    return 0;
  }
  if ((d->ldc & 7) || (d->N & 7) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return 0;   // the staged bf16 epilogue
  if (d->preact != nullptr || (d->act != 0 && d->act != 3)) return 0;
  if (d->residual != nullptr && d->colscale == nullptr) return 0;    // a residual only as part of the eval fold
  if (bn == 192) return launch_halo_fs2<192>(p, st);
  if (bn == 128) return launch_halo_fs2<128>(p, st);
  return 0;
}


// Merged strided dgrad (HtrvtGemmDesc.cls_h == -2): 1 launched, 0 not served, < 0 error.  Sets p.tiles_m / tiles_n / Hq / Wq.
// `probe`: only answer (tiles_m, or 0) -- htrvt_gemm_dgrad_merged_tiles
int gemm_halo_s2_try_launch(const HtrvtGemmDesc* d, KParams& p, hipStream_t st, bool probe) {
  if (d->gather != HTRVT_GATHER_CONV_DGRAD || d->dtype != HTRVT_BF16 || d->cls_h != -2) return 0;
  if (d->tile != 0 && d->tile != 4 && d->tile != 12) return 0;
  if (d->kh != 3 || d->kw != 3 || d->ph != 1 || d->pw != 1 || d->sh != 2 || (d->sw != 1 && d->sw != 2)) return 0;
  if ((d->Hi & 1) || d->Wi % d->sw || d->Ho != d->Hi / 2 || d->Wo != d->Wi / d->sw || (d->Wo % 256) != 0) return 0;
  if (d->batch > 1 || d->split_k > 1 || d->c_f32 || d->N != d->Ci || d->M != d->nB * d->Hi * d->Wi) return 0;
  if (d->K != (9 + (d->A2 != nullptr ? 1 : 0)) * d->Cpad) return 0;
  if ((d->ldc & 7) || (d->N & 7) || (d->Co & 7) || (reinterpret_cast<unsigned long long>(d->C) & 15)) return 0;
  if (d->colscale != nullptr || d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->relu_scale != nullptr) return 0;
  const long long lim = (1ll << 31) - 64;
  const long long abytes = (long long)d->nB * d->Ho * d->Wo * d->Co * 2;
  if (abytes >= lim || (long long)d->N * d->ldb * 2 >= lim) return 0;
  if (d->A2 != nullptr) {
    const long long diff = (const char*)d->A2 - (const char*)d->A;
    if (diff <= 0 || diff + abytes >= lim) return 0;
  }
  const int bn = d->N <= 128 ? 128 : 192;
  p.Hq = d->Hi / 2;
  p.Wq = d->Wi / d->sw;
  p.tiles_m = 2 * d->sw * d->nB * p.Hq * (p.Wq / 256);
  p.tiles_n = (d->N + bn - 1) / bn;
  if (probe) return p.tiles_m;
  p.cls_h = p.cls_w = -1;
  p.extra_off = d->A2 != nullptr ? (unsigned)((const char*)d->A2 - (const char*)d->A) : 0u;
  return bn == 192 ? launch_halo_s2<192>(p, st) : launch_halo_s2<128>(p, st);
}

}  // namespace htrvt

#ifdef HTRVT_EXP_STAMP
extern "C" int htrvt_debug_read_halo(void* dst, int nbytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(htrvt_dbg), nbytes, 0, hipMemcpyDeviceToHost);
}
#endif
