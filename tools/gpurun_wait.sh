#!/bin/bash
# Dev-container helper: submit ONE gpurun call, re-submitting only while the pool answers "no box / slot free"
# (exit code 3: nothing ran, nothing was charged).  Any other exit code -- success, test failure, timeout -- ends it.
#   tools/gpurun_wait.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
