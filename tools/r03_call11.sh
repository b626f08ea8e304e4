#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "wgrad" > $O/t_hw.log 2>&1; rc=$?; echo "hwgrad tests rc=$rc"; tail -8 $O/t_hw.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_attention_gpu.py -m gpu -x -q > $O/t_hw2.log 2>&1; echo "gemm+attn tests rc=$?"; tail -3 $O/t_hw2.log
timeout -k 10 600 python tools/bench_gemm.py --only s1conv --tiles 3 13 --rounds 3 > $O/bg_hwgrad.txt 2>&1; echo "rc=$?"; grep wgrad $O/bg_hwgrad.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table11.txt > $O/b128_11.json 2> $O/b128_11.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_11.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
