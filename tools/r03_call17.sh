#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/t17.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t17.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table17.txt > $O/b128_17.json 2> $O/b128_17.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_17.json').read().strip().splitlines()[-1]);print('relu-from-bn',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
HTRVT_ENGINE_OVERRIDE=relu_mask_from_bn=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table17_off.txt > $O/b128_17_off.json 2> $O/b128_17_off.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_17_off.json').read().strip().splitlines()[-1]);print('relu-src    ',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
done
grep "gemm_halo_kernel<192, true>" $O/gemm_table17.txt $O/gemm_table17_off.txt
