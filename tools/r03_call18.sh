#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python tools/bench_gemm.py --only lwgrad --tiles 3 4 6 --rounds 2 > $O/bg_lwgrad.txt 2>&1; echo "rc=$?"; cat $O/bg_lwgrad.txt | tail -40
