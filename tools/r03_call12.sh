#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_aux_gpu.py tests/test_determinism_gpu.py tests/test_dp_gpu.py -m gpu -x -q > $O/t12.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/t12.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table12.txt > $O/b128_12.json 2> $O/b128_12.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_12.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t12_all.log 2>&1; echo "all tests rc=$?"; tail -4 $O/t12_all.log
