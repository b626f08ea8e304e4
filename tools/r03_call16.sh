#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm8p_gpu.py tests/test_model_gpu.py tests/test_stem_gpu.py -m gpu -x -q > $O/t16.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t16.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --forward-only --no-cpu-baseline --gemm-table $O/gemm_table16_fwd.txt > $O/fwd16.json 2> $O/fwd16.err; echo "fwd rc=$?"; python -c "
import json;d=json.loads(open('$O/fwd16.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'])"
HTRVT_GEMM_NOHALO_EVAL=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path > $O/b128_16.json 2> $O/b128_16.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_16.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"
