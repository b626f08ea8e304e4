// gemm_dma_bn128.hip -- instantiations of the LDS-DMA GEMM kernel for N tiles of 128 columns (own translation
// unit: hipcc spends ~1 minute per dozen kernel variants, the Makefile builds the units in parallel)
#include "gemm_dma_impl.h"

namespace htrvt {
int gemm_dma_dispatch_bn128(const HtrvtGemmDesc* d, const KParams& p, int zdim, hipStream_t st, bool spec) {
  if (spec) return dispatch<256, 128, 1>(d, p, zdim, st);
  return dispatch<256, 128, 0>(d, p, zdim, st);
}
// three LDS stages (144 KB): two k-tiles of DMA in flight
int gemm_dma_dispatch_bn128_s3(const HtrvtGemmDesc* d, const KParams& p, int zdim, hipStream_t st, bool spec) {
  if (spec) return dispatch<256, 128, 1, 3>(d, p, zdim, st);
  return dispatch<256, 128, 0, 3>(d, p, zdim, st);
}
}  // namespace htrvt
