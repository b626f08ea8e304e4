#!/bin/bash
# Hardware-counter passes over one bench.py training step (run on the GPU box through gpurun).  Counters are collected
# in their own runs, with the kernel trace only (the pool forbids PMC together with the API / memory-copy traces);
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots), MI355X_MICROARCH.md "rocprofv3 PMC slots".
#   tools/collect_pmc.sh <out-dir-under-gpurun_out> [bench args...]
OUT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}/gpurun_out/$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BENCH="$ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-path --no-overlap-wgrad $*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters...
  local name=$1; shift
  if rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 $BENCH > "$OUT/$name.log" 2>&1; then
    echo "pass $name done: $(ls "$OUT/$name"/*/ 2>/dev/null | tr '\n' ' ')"
  else
    echo "pass $name FAILED: $(tail -3 "$OUT/$name.log")"
  fi
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
