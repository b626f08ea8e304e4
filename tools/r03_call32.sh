#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py tests/test_variants_gpu.py -m gpu -x -q > $O/t32.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t32.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/bench_attn.py > $O/bench_attn32.txt 2>&1; tail -8 $O/bench_attn32.txt
