#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_gemm8p_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/t26.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/t26.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_b16_26.txt > $O/b16_26.json 2> $O/b16_26.err; python -c "
import json;d=json.loads(open('$O/b16_26.json').read().strip().splitlines()[-1]);print(16, d['ms_per_step'],d['value'],d['roofline']['all_mfma_tflops'])"; grep "8192, 768, 6912\|32768, 384, 3456" $O/gemm_b16_26.txt
