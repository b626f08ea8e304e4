#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py -m gpu -x -q > $O/t19.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t19.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table19.txt > $O/b128_19.json 2> $O/b128_19.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_19.json').read().strip().splitlines()[-1]);print('raw sums',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
HTRVT_LIB=$PWD/htr-vt_amd/lib/exp_prev.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table19_prev.txt > $O/b128_19_prev.json 2> $O/b128_19_prev.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_19_prev.json').read().strip().splitlines()[-1]);print('previous',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
done
grep "gemm_halo_kernel<192, true>\|0, 0, 2, 1048576\|0, 0, 2, 262144\|0, 0, 2, 65536" $O/gemm_table19.txt $O/gemm_table19_prev.txt
