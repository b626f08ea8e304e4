#!/bin/bash
# round-3 GPU call 1: full GPU test suite, parity log, small-batch lines, BASELINE config lines
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python -m pytest tests -m gpu -x -q > $O/tests1.log 2>&1; echo "tests rc=$?" | tee -a $O/tests1.log
tail -5 $O/tests1.log
python -m pytest tests/test_full_shape_gpu.py tests/test_model_gpu.py tests/test_train_iter_gpu.py -m gpu -s -q > $O/parity.log 2>&1; echo "parity rc=$?"
python bench.py --steps 10 --warmup 3 > $O/b128.json 2> $O/b128.err; tail -c 600 $O/b128.json
for b in 64 32 16; do python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_b$b.txt > $O/b$b.json 2> $O/b$b.err; echo "b$b rc=$?"; done
python bench.py --forward-only --no-cpu-baseline > $O/cfg2_fwd.json 2> $O/cfg2_fwd.err; echo "fwd rc=$?"
python bench.py --width 2048 --batch 64 --no-cpu-baseline --no-parity-path > $O/cfg4_w2048.json 2> $O/cfg4.err; echo "w2048 rc=$?"
python bench.py --embed-dim 512 --depth 12 --heads 8 --nb-cls 90 --batch 32 --no-cpu-baseline --no-parity-path > $O/cfg5_bf16.json 2> $O/cfg5_bf16.err; echo "cfg5 bf16 rc=$?"
python bench.py --embed-dim 512 --depth 12 --heads 8 --nb-cls 90 --batch 32 --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline > $O/cfg5_f32.json 2> $O/cfg5_f32.err; echo "cfg5 f32 rc=$?"
python bench.py --sam --no-cpu-baseline --no-parity-path > $O/sam.json 2> $O/sam.err; echo "sam rc=$?"
