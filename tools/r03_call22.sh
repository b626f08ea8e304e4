#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "halo" > $O/t22.log 2>&1; echo "tests rc=$?"; tail -3 $O/t22.log
timeout -k 10 300 python tools/bench_gemm.py --only s1conv --rounds 3 --libs htr-vt_amd/lib/exp_before.so htr-vt_amd/lib/libhtrvt_hip.so > $O/bg_wait.txt 2>&1; echo "rc=$?"; grep -v wgrad $O/bg_wait.txt | tail -16
