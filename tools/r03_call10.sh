#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "downsample or fused or gathers" > $O/t_ds.log 2>&1; rc=$?; echo "ds tests rc=$rc"; tail -8 $O/t_ds.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests10.log 2>&1; echo "all tests rc=$?"; tail -4 $O/tests10.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table10.txt > $O/b128_10.json 2> $O/b128_10.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_10.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
python -m pytest tests/test_full_shape_gpu.py -m gpu -x -q -s > $O/t_full.log 2>&1; echo "full-shape rc=$?"; grep -E "oracle|rel-L2|worst|passed|failed" $O/t_full.log | tail -14
