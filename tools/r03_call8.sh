#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py tests/test_variants_gpu.py -m gpu -x -q > $O/t_attn.log 2>&1; rc=$?; echo "attn tests rc=$rc"; tail -12 $O/t_attn.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_full_shape_gpu.py -m gpu -x -q > $O/t_model.log 2>&1; echo "model tests rc=$?"; tail -4 $O/t_model.log
timeout -k 10 300 python tools/bench_attn.py > $O/bench_attn.txt 2>&1; cat $O/bench_attn.txt
