#!/bin/bash
# The round's GPU evidence in one call (run through gpurun): test suite, parity prints, bench lines for every BASELINE
# configuration and for the per-rank batches of a strong-scaling run, rocprofv3 kernel-trace statistics of the bench
# command and the four hardware-counter passes.  Summaries are written by tools/make_round_docs.py / prof_summary.py /
# pmc_summary.py from what this leaves under gpurun_out/<tag>/.
#   tools/round_evidence.sh <tag> [tests|bench|ab|prof|quick]     (a gpurun call is limited to 20 minutes: one part per call)
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
PART=${2:-all}
if [ "$PART" = "all" ] || [ "$PART" = "tests" ]; then
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests1.log 2>&1; echo "tests rc=$?" | tee -a $O/tests1.log; tail -3 $O/tests1.log
timeout -k 10 900 python -m pytest tests/test_full_shape_gpu.py tests/test_model_gpu.py tests/test_train_iter_gpu.py tests/test_graph_gpu.py -m gpu -s -q > $O/parity.log 2>&1; echo "parity rc=$?"
fi
if [ "$PART" = "tests" ]; then exit 0; fi
if [ "$PART" != "prof" ] && [ "$PART" != "ab" ]; then
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --gemm-table $O/gemm_table.txt > $O/b128.json 2> $O/b128.err; echo "b128 rc=$?"; tail -c 400 $O/b128.json
for b in 64 32 16; do timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_b$b.txt > $O/b$b.json 2> $O/b$b.err; echo "b$b rc=$?"; done
timeout -k 10 300 python bench.py --forward-only --no-cpu-baseline > $O/cfg2_fwd.json 2> $O/cfg2_fwd.err; echo "fwd rc=$?"
timeout -k 10 300 python bench.py --width 2048 --batch 64 --no-cpu-baseline --no-parity-path > $O/cfg4_w2048.json 2> $O/cfg4.err; echo "w2048 rc=$?"
timeout -k 10 300 python bench.py --embed-dim 512 --depth 12 --heads 8 --nb-cls 90 --batch 32 --no-cpu-baseline --no-parity-path > $O/cfg5_bf16.json 2> $O/cfg5_bf16.err; echo "cfg5 bf16 rc=$?"
timeout -k 10 300 python bench.py --embed-dim 512 --depth 12 --heads 8 --nb-cls 90 --batch 32 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline > $O/cfg5_f32.json 2> $O/cfg5_f32.err; echo "cfg5 f32 rc=$?"
timeout -k 10 300 python bench.py --sam --no-cpu-baseline --no-parity-path > $O/sam.json 2> $O/sam.err; echo "sam rc=$?"
timeout -k 10 300 python bench.py --dtype split_bf16 --steps 3 --warmup 3 --no-cpu-baseline --no-parity-path > $O/split_bf16.json 2> $O/split_bf16.err; echo "split rc=$?"
for b in 128 16; do timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --graph on > $O/graph_b$b.json 2> $O/graph_b$b.err; echo "graph b$b rc=$?"; done
fi
if [ "$PART" = "ab" ]; then      # same-box A/B against the previous round's tree (tools/r05_ab.sh) + the convolution / encoder micro-benchmarks
bash tools/r05_ab.sh $TAG
timeout -k 10 300 python tools/bench_gemm.py --only "s1conv sconv c1x1" --rounds 2 > $O/conv_table.log 2>&1; echo "conv table rc=$?"
timeout -k 10 300 python tools/bench_gemm.py --only enc --tiles 9 0 --rounds 2 > $O/enc_ab.log 2>&1; echo "enc ab rc=$?"
exit 0
fi
if [ "$PART" = "quick" ] || [ "$PART" = "bench" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-path --no-overlap-wgrad > $O/prof.log 2>&1; echo "prof rc=$?"
cd $ROOT
timeout -k 10 1000 bash tools/collect_pmc.sh $TAG/pmc; echo "pmc rc=$?"
