#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
L=htr-vt_amd/lib
timeout -k 10 600 python tools/bench_gemm.py --only s1conv --libs $L/libhtrvt_hip.so $L/exp_nodma.so $L/exp_nomma.so --rounds 2 > $O/bg_exp_conv.txt 2>&1; echo "rc=$?"; cat $O/bg_exp_conv.txt
timeout -k 10 600 python tools/bench_gemm.py --only plain --libs $L/libhtrvt_hip.so $L/exp_nodma.so $L/exp_nomma.so --rounds 2 > $O/bg_exp_plain.txt 2>&1; echo "rc=$?"; cat $O/bg_exp_plain.txt
