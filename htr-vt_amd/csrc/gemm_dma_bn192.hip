// gemm_dma_bn192.hip -- instantiations of the LDS-DMA GEMM kernel for N tiles of 192 columns (own translation
// unit: hipcc spends ~1 minute per dozen kernel variants, the Makefile builds the units in parallel)
#include "gemm_dma_impl.h"

namespace htrvt {
int gemm_dma_dispatch_bn192(const HtrvtGemmDesc* d, const KParams& p, int zdim, hipStream_t st, bool spec) {
  if (spec) return dispatch<256, 192, 1>(d, p, zdim, st);
  return dispatch<256, 192, 0>(d, p, zdim, st);
}
}  // namespace htrvt

#ifdef HTRVT_EXP_STAMP
extern "C" int htrvt_debug_read_bn192(void* dst, int nbytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(htrvt_dbg), nbytes, 0, hipMemcpyDeviceToHost);
}
#endif
