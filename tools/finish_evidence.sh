#!/bin/bash
# Dev container, after `gpurun -- bash tools/round_evidence.sh <tag>`: turn gpurun_out/<tag>/ into the committed summaries.
#   tools/finish_evidence.sh <tag> [round]      e.g. tools/finish_evidence.sh r04a r04
TAG=${1:?tag}; RND=${2:-r05}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
python tools/make_round_docs.py gpurun_out/$TAG $RND || exit 1
python tools/prof_summary.py gpurun_out/$TAG/prof profiles/${RND}_step_bf16_b128.md --steps 9 \
  --title "Round ${RND#r0}: HTR-VT training step, bf16, B=128, 64x1024 (rocprofv3 kernel-trace stats)" \
  --cmd "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-path --no-overlap-wgrad   (weight gradients kept on the main stream so that kernels do not overlap in the trace; 5 timed + 2 warm-up + 2 event-profiled steps)" | tail -1
python tools/pmc_summary.py gpurun_out/$TAG/pmc profiles/${RND}_pmc --steps 4 | tail -1
cp gpurun_out/$TAG/gemm_table.txt profiles/${RND}_gemm_table.txt
