#!/bin/bash
# Same-box A/B of the round: this tree against the tree of the previous round's last commit (its sources + library copied to
# _ab_r04/ by hand: `git worktree add /tmp/r04 a4d39f9 && make -C /tmp/r04/htr-vt_amd/csrc`), interleaved, 3 x 2 runs of
# bench.py --steps 20 --warmup 5.  The boxes of the pool differ by up to 5-7 % on the MFMA kernels: only runs on ONE box compare.
#   tools/r05_ab.sh <tag>      -> gpurun_out/<tag>/ab.txt
TAG=${1:-r05ab}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
: > $O/ab.txt
for rep in 1 2 3; do
  for mode in r05 r04; do
    if [ $mode = r04 ]; then cd $ROOT/_ab_r04; else cd $ROOT; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${mode}_$rep.json 2> $O/ab_${mode}_$rep.err
    python - $O/ab_${mode}_$rep.json $mode $rep >> $O/ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print(f"{sys.argv[2]:4s} run {sys.argv[3]}: {d['ms_per_step']:.3f} ms/step  {d['value']:.1f} images/s  dominant symbol {r['kernel']} {r['achieved']} TFLOP/s (frac {r['frac']})  all MFMA launches {r['all_mfma_tflops']} TFLOP/s over {r['mfma_ms_per_step']} ms")
PY
  done
done
cat $O/ab.txt
