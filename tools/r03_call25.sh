#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 300 python bench.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_b16_25.txt > $O/b16_25.json 2> $O/b16_25.err; grep "4096, 768\|4096, 2304\|4096, 3072" $O/gemm_b16_25.txt
