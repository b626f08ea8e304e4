#!/bin/bash
# Same-box A/B of the round's switchable changes (persistent + MN-major 8-phase GEMMs: env switches of csrc/gemm8p.hip; ReLU bit mask: engine switch) against the build without them, interleaved:
# the boxes of the pool differ by up to 5 % on the MFMA kernels, so only runs on ONE box compare.
#   tools/r04_ab.sh <tag>      -> gpurun_out/<tag>/ab.txt
TAG=${1:-r04ab}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
: > $O/ab.txt
for rep in 1 2 3; do
  for mode in new old; do
    if [ $mode = old ]; then export HTRVT_NO_PERSISTENT_GEMM=1 HTRVT_NO_MNMAJOR_8PHASE=1 HTRVT_ENGINE_OVERRIDE="relu_bitmask=0"; else unset HTRVT_NO_PERSISTENT_GEMM HTRVT_NO_MNMAJOR_8PHASE HTRVT_ENGINE_OVERRIDE; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${mode}_$rep.json 2> $O/ab_${mode}_$rep.err
    python - $O/ab_${mode}_$rep.json $mode $rep >> $O/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:4s} run {sys.argv[3]}: {d['ms_per_step']:.3f} ms/step  {d['value']:.1f} images/s  all MFMA launches {d['roofline']['all_mfma_tflops']} TFLOP/s over {d['roofline']['mfma_ms_per_step']} ms")
PY
  done
done
cat $O/ab.txt
