#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
L=htr-vt_amd/lib
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_determinism_gpu.py -m gpu -x -q > $O/t_stagger.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t_stagger.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python tools/bench_gemm.py --only s1conv --libs $L/libhtrvt_hip.so $L/exp_nostagger.so --rounds 3 > $O/bg_stagger.txt 2>&1; echo "rc=$?"; grep wgrad $O/bg_stagger.txt
timeout -k 10 600 python tools/bench_gemm.py --only plain --libs $L/libhtrvt_hip.so $L/exp_nostagger.so --rounds 3 > $O/bg_stagger_plain.txt 2>&1; echo "rc=$?"; grep TN $O/bg_stagger_plain.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table6.txt > $O/b128_6.json 2> $O/b128_6.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_6.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
