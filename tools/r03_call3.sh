#!/bin/bash
# round-3 GPU call 3: halo-staged conv kernel -- correctness, then A/B against the generic gather (tile 5)
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py -m gpu -x -q > $O/t_halo.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -15 $O/t_halo.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python tools/bench_gemm.py --only s1conv --tiles 5 12 --rounds 3 > $O/bg_halo.txt 2>&1; echo "halo bench rc=$?"; cat $O/bg_halo.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests3.log 2>&1; echo "all tests rc=$?"; tail -5 $O/tests3.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table3.txt > $O/b128_3.json 2> $O/b128_3.err; echo "bench rc=$?"; tail -c 1200 $O/b128_3.json
