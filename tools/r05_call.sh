#!/bin/bash
# one gpurun call of round 5: tools/r05_call.sh <tag> '<commands...>' (runs from the repo root, output under gpurun_out/<tag>/)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
export O
eval "$@"
