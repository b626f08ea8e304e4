#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_gemm8p_gpu.py -m gpu -x -q -k "fused or halo or dgrad" > $O/t34.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t34.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table34.txt > $O/b128_34.json 2> $O/b128_34.err; python -c "
import json;d=json.loads(open('$O/b128_34.json').read().strip().splitlines()[-1]);print('new ',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
HTRVT_LIB=$PWD/htr-vt_amd/lib/exp_prev.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table34_prev.txt > $O/b128_34_prev.json 2> $O/b128_34_prev.err; python -c "
import json;d=json.loads(open('$O/b128_34_prev.json').read().strip().splitlines()[-1]);print('prev',d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
done
grep "gemm_halo_kernel<192, true>" $O/gemm_table34.txt $O/gemm_table34_prev.txt
