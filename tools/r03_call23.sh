#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python tools/bench_gemm.py --only encsmall --tiles 0 1 4 7 8 --rounds 2 > $O/bg_encsmall.txt 2>&1; echo "rc=$?"; tail -16 $O/bg_encsmall.txt
