#!/bin/bash
# round-3 GPU call 2: the 8-phase GEMM family -- correctness first, then A/B against the LDS-DMA family
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q > $O/t_gemm8p.log 2>&1; rc=$?; echo "gemm8p tests rc=$rc"; tail -15 $O/t_gemm8p.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python tools/bench_gemm.py --only enc --tiles 4 10 11 --rounds 3 > $O/bg_enc.txt 2>&1; echo "enc rc=$?"; cat $O/bg_enc.txt
timeout -k 10 600 python tools/bench_gemm.py --only conv --tiles 4 10 11 --rounds 2 > $O/bg_conv.txt 2>&1; echo "conv rc=$?"; cat $O/bg_conv.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests2.log 2>&1; echo "all tests rc=$?"; tail -5 $O/tests2.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table2.txt > $O/b128_2.json 2> $O/b128_2.err; echo "bench rc=$?"; tail -c 1500 $O/b128_2.json
