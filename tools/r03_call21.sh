#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python tools/bench_gemm.py --only s1conv --rounds 3 --libs htr-vt_amd/lib/libhtrvt_hip.so htr-vt_amd/lib/exp_regstage.so > $O/bg_regstage.txt 2>&1; echo "rc=$?"; grep -v wgrad $O/bg_regstage.txt | tail -20
