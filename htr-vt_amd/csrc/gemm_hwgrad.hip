// gemm_hwgrad.hip -- instantiations + host-side eligibility of the halo-staged 3x3 (W stride 1) conv weight-gradient kernels
// (gemm_hwgrad_impl.h); called from gemm_dma_try_launch before the generic MN-major gather kernel.
#include "gemm_hwgrad_impl.h"

namespace htrvt {

// channel chunk of the tile: 128 where it divides the padded channel count; else (a multiple of 64 only: layer 1's 192)
// the 128-wide tile made of two independent 64-channel units (PAIR), or plain 64-channel tiles when asked for (tile 17: A/B)
// round 5: a padded channel count that is a multiple of 96 but not of 128 (layer 1: 192) takes 96-channel units on the 16x16x32
// kernel (gemm_hwgrad16_kernel: 288 x 192 tiles, no half-empty pair); tile 18 asks for the paired form of round 4 (A/B)
int gemm_hwgrad_bn(const HtrvtGemmDesc* d);
// (the 16x16x32 kernel with 128-channel units at layers 2-3 -- 96 accumulators + two k-steps of fragments, 25 registers in scratch --
// measured EQUAL to the 32x32x16 kernel there: 0.584 / 0.572 against 0.587 / 0.572 ms, profiles/r05_experiments.md (i); not kept)
bool gemm_hwgrad_is16(const HtrvtGemmDesc* d) {
  if (gemm_hwgrad_bn(d) != 192) return false;
  // column stride 2 (conv1 of layer2.0 / 3.0): odd / even pixel images of x, gemm_hwgrad16_kernel<96, 192, 2> -- exact, but MEASURED
  // SLOWER than the generic gather at its best split (0.422 against 0.410 ms at layer2.0, 0.413 against 0.367 at layer3.0:
  // 27 + 24 DMA pieces per 288 x 192 k-tile; profiles/r05_experiments.md (j)), so only on request (tile 13: tests, A/B runs)
  if (d->sw == 2) return d->Cpad % 96 == 0 && d->tile == 13;
  return d->Cpad % 128 != 0 && d->Cpad % 96 == 0 && (d->tile == 0 || d->tile == 13);
}
int gemm_hwgrad_cc(const HtrvtGemmDesc* d) { return gemm_hwgrad_is16(d) ? 96 : ((d->Cpad % 128 == 0 || d->tile != 17) ? 128 : 64); }
bool gemm_hwgrad_pair(const HtrvtGemmDesc* d) { return !gemm_hwgrad_is16(d) && d->Cpad % 128 != 0 && d->tile != 17; }
int gemm_hwgrad_bn(const HtrvtGemmDesc* d) {
  const int p192 = (d->N + 191) / 192 * 192, p128 = (d->N + 127) / 128 * 128;
  return p192 <= p128 ? 192 : 128;
}

bool gemm_hwgrad_serves(const HtrvtGemmDesc* d) {
  if (d->gather != HTRVT_GATHER_CONV_WGRAD || d->dtype != HTRVT_BF16) return false;
  if (d->tile != 0 && d->tile != 13 && d->tile != 17 && d->tile != 18) return false;      // 13: this kernel where eligible (17: without unit pairing); 3 / 4 / 6: the generic kernels (A/B)
  if (d->kh != 3 || d->kw != 3 || (d->sh != 1 && d->sh != 2) || (d->sw != 1 && d->sw != 2) || d->ph != 1 || d->pw != 1) return false;   // row stride 2: layer1.0.conv1
  if (d->Ho != (d->Hi - 1) / d->sh + 1 || (d->Wo % 64) != 0) return false;     // a k-tile = 64 pixels of one output row
  if (d->sw == 1 ? d->Wo != d->Wi : ((d->Wi & 1) || d->Wo != d->Wi / 2 || !gemm_hwgrad_is16(d))) return false;   // column stride 2: conv1 of layer2.0 / 3.0, odd / even pixel images
  if (!d->c_f32 || d->batch > 1 || d->Cpad % 64 != 0 || d->M != 9 * d->Cpad || d->N != d->Co) return false;
  if (d->bias != nullptr || d->act != 0 || d->preact != nullptr || d->residual != nullptr || d->colstats != nullptr) return false;
  const long long lim = (1ll << 31) - 64;
  if ((long long)d->nB * d->Hi * d->Wi * d->Ci * 2 >= lim || (long long)d->K * d->ldb * 2 >= lim) return false;
  return true;
}

// work groups per pixel range (for the split-K heuristic of the caller)
int gemm_hwgrad_tiles(const HtrvtGemmDesc* d) {
  const int tm = gemm_hwgrad_pair(d) ? (3 * (d->Cpad / 64) + 1) / 2 : 3 * (d->Cpad / gemm_hwgrad_cc(d));
  return tm * ((d->N + gemm_hwgrad_bn(d) - 1) / gemm_hwgrad_bn(d));
}

int gemm_hwgrad_try_launch(const HtrvtGemmDesc* d, KParams& p, int zdim, hipStream_t st) {
  if (!gemm_hwgrad_serves(d)) return 0;
  const int cc = gemm_hwgrad_cc(d), bn = gemm_hwgrad_bn(d);
  p.tiles_n = (d->N + bn - 1) / bn;
  if (gemm_hwgrad_is16(d)) {
    p.tiles_m = 3 * (d->Cpad / cc);
    return d->sw == 2 ? launch_hwgrad16<96, 192, 2>(p, zdim, st) : launch_hwgrad16<96, 192>(p, zdim, st);
  }
  if (gemm_hwgrad_pair(d)) {
    p.tiles_m = (3 * (d->Cpad / 64) + 1) / 2;
    return bn == 192 ? launch_hwgrad<128, 192, true>(p, zdim, st) : launch_hwgrad<128, 128, true>(p, zdim, st);
  }
  p.tiles_m = 3 * (d->Cpad / cc);
  if (cc == 128 && bn == 192) return launch_hwgrad<128, 192>(p, zdim, st);
  if (cc == 128 && bn == 128) return launch_hwgrad<128, 128>(p, zdim, st);
  if (cc == 64 && bn == 192) return launch_hwgrad<64, 192>(p, zdim, st);
  return launch_hwgrad<64, 128>(p, zdim, st);
}

}  // namespace htrvt

// host query for the split-K heuristic of the caller: workgroups per K range, tile rows, tile columns of the launch the
// halo-staged weight-gradient kernel would make for this descriptor; 0 when the generic kernels serve it
extern "C" int htrvt_gemm_wgrad_tiling(const HtrvtGemmDesc* d, int* tile_rows, int* tile_cols) {
  if (d == nullptr) return 0;
  if (d->gather == HTRVT_GATHER_NONE) {      // a Linear weight gradient dW = dy^T x: the MN-major 8-phase kernel's 256 x 256 tiles (gemm8p.hip)
    if (!htrvt::gemm8pt_serves(d) || !(d->split_k > 1 ? d->accumulate != 0 : d->accumulate == 0)) return 0;
    if (tile_rows) *tile_rows = 256;
    if (tile_cols) *tile_cols = 256;
    return ((d->M + 255) / 256) * ((d->N + 255) / 256);
  }
  if (!htrvt::gemm_hwgrad_serves(d)) return 0;
  if (tile_rows) *tile_rows = 3 * htrvt::gemm_hwgrad_cc(d);
  if (tile_cols) *tile_cols = htrvt::gemm_hwgrad_bn(d);
  return htrvt::gemm_hwgrad_tiles(d);
}
