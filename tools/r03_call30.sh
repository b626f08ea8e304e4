#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for b in 1 4 8; do timeout -k 10 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/b${b}_30.json 2> $O/b${b}_30.err; python -c "
import json;d=json.loads(open('$O/b${b}_30.json').read().strip().splitlines()[-1]);print($b, d['ms_per_step'],d['value'],d['roofline']['mfma_ms_per_step'])"; done
