#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -m gpu -x -q > $O/t_attn2.log 2>&1; rc=$?; echo "attn tests rc=$rc"; tail -3 $O/t_attn2.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/bench_attn.py > $O/bench_attn2.txt 2>&1; cat $O/bench_attn2.txt
timeout -k 10 300 python tools/bench_gemm.py --only plain --tiles 0 6 --rounds 3 > $O/bg_tn256.txt 2>&1; grep TN $O/bg_tn256.txt
