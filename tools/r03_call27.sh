#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for b in 16 32; do for det in "" "--deterministic"; do
timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path $det > $O/b${b}_27.json 2> $O/b${b}_27.err; python -c "
import json;d=json.loads(open('$O/b${b}_27.json').read().strip().splitlines()[-1]);print($b, '$det', d['ms_per_step'],d['value'],d['roofline']['all_mfma_tflops'])"; done; done
