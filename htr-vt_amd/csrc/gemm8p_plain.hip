// gemm8p_plain.hip -- instantiations of the 8-phase GEMM kernel for plain (Linear) operands: forward (bias, GELU +
// saved pre-activation, residual) and dgrad against the transposed weight copy (plain, * GELU').  Own translation unit
// (the Makefile builds the units in parallel).
#include "gemm8pp_impl.h"
#include "gemm8pt_impl.h"

namespace htrvt {

template <class C>
static int by_epi(int epi, const KParams& p, int zdim, hipStream_t st) {
  using namespace g8;
  switch (epi) {
    case 0: return launch<C, 0, 0>(p, zdim, st);
    case E_RES: return launch<C, 0, E_RES>(p, zdim, st);
    case E_GELU: return launch<C, 0, E_GELU>(p, zdim, st);
    case E_GELUGRAD: return launch<C, 0, E_GELUGRAD>(p, zdim, st);
    case E_F32: return launch<C, 0, E_F32>(p, zdim, st);      // float32 C (+ bias): the Linear products of the split-bf16 parity path
    default: return 0;
  }
}

// persistent form (gemm8pp_impl.h): bias / bias + GELU epilogues; nwg workgroups walk all tiles
int gemm8pp_dispatch_plain(int bn, int epi, const KParams& p, int nwg, hipStream_t st) {
  using namespace g8;
  if (bn == 256) {
    if (epi == 0) return launch_persistent<Cfg<256, 2, 4>, 0>(p, nwg, st);
    if (epi == E_GELU) return launch_persistent<Cfg<256, 2, 4>, E_GELU>(p, nwg, st);
  } else {
    if (epi == 0) return launch_persistent<Cfg<192, 4, 2>, 0>(p, nwg, st);
    if (epi == E_GELU) return launch_persistent<Cfg<192, 4, 2>, E_GELU>(p, nwg, st);
    if (epi == E_RES) return launch_persistent<Cfg<192, 4, 2>, E_RES>(p, nwg, st);
  }
  return 0;
}

// MN-major x MN-major operands, float32 output (gemm8pt_impl.h): the Linear weight gradients
int gemm8pt_dispatch(const KParams& p, int zdim, hipStream_t st) { return g8::launch_t(p, zdim, st); }

int gemm8p_dispatch_plain(int bn, int epi, const KParams& p, int zdim, hipStream_t st) {
  if (bn == 256) return by_epi<g8::Cfg<256, 2, 4>>(epi, p, zdim, st);
  return by_epi<g8::Cfg<192, 4, 2>>(epi, p, zdim, st);
}

}  // namespace htrvt
