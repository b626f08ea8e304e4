// gemm8p_conv.hip -- instantiations of the 8-phase kernel for the implicit-GEMM convolutions: forward (train: BatchNorm
// column sums; eval: folded BatchNorm scale / shift / residual / ReLU) and dgrad (plain, + residual, + backward-of-ReLU
// mask and one or two BatchNorm-backward sum sets).
#include "gemm8p_impl.h"

namespace htrvt {

template <class C>
static int conv_by_epi(int gather, int epi, const KParams& p, int zdim, hipStream_t st) {
  using namespace g8;
  if (gather == HTRVT_GATHER_CONV_FWD) {
    switch (epi) {
      case 0: return launch<C, 1, 0>(p, zdim, st);
      case E_CSTATS: return launch<C, 1, E_CSTATS>(p, zdim, st);
      case E_SCALE_RELU: return launch<C, 1, E_SCALE_RELU>(p, zdim, st);
      case E_SCALE_RELU | E_RES: return launch<C, 1, E_SCALE_RELU | E_RES>(p, zdim, st);
      default: return 0;
    }
  }
  switch (epi) {
    case 0: return launch<C, 2, 0>(p, zdim, st);
    case E_RES: return launch<C, 2, E_RES>(p, zdim, st);
    case E_RELUMASK | E_BNB1: return launch<C, 2, E_RELUMASK | E_BNB1>(p, zdim, st);
    case E_RES | E_RELUMASK | E_BNB1: return launch<C, 2, E_RES | E_RELUMASK | E_BNB1>(p, zdim, st);
    case E_RES | E_RELUMASK | E_BNB1 | E_BNB2:
      // two BatchNorm sum sets beside 128 accumulators do not fit 256 registers (the 256-column tile spills); the case
      // arises at the first block of a stage, whose gradient has 192 / 384 channels = 192-column tiles
      if constexpr (C::BN == 256) return 0;
      else return launch<C, 2, E_RES | E_RELUMASK | E_BNB1 | E_BNB2>(p, zdim, st);
    default: return 0;
  }
}

int gemm8p_dispatch_conv(int bn, int gather, int epi, const KParams& p, int zdim, hipStream_t st) {
  if (bn == 256) return conv_by_epi<g8::Cfg<256, 2, 4>>(gather, epi, p, zdim, st);
  return conv_by_epi<g8::Cfg<192, 4, 2>>(gather, epi, p, zdim, st);
}

}  // namespace htrvt
