#!/bin/bash
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_variants_gpu.py -m gpu -x -q -s > $O/t_variants.log 2>&1; echo "variants rc=$?"; tail -8 $O/t_variants.log
timeout -k 10 600 python tools/bench_gemm.py --only conv --tiles 3 4 --rounds 2 > $O/bg_wgrad_spec.txt 2>&1; echo "rc=$?"; grep wgrad $O/bg_wgrad_spec.txt
timeout -k 10 600 python tools/bench_gemm.py --only plain --tiles 3 4 --rounds 2 > $O/bg_plain_spec.txt 2>&1; echo "rc=$?"; cat $O/bg_plain_spec.txt
