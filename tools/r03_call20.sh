#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/t20.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t20.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table20.txt > $O/b128_20.json 2> $O/b128_20.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_20.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
grep "1048576, 192, 1728\|1728, 192, 1048576" $O/gemm_table20.txt
