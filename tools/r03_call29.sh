#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t29.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t29.log
if [ $rc -ne 0 ]; then exit 1; fi
for b in 16 32 128; do timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path > $O/b${b}_29.json 2> $O/b${b}_29.err; python -c "
import json;d=json.loads(open('$O/b${b}_29.json').read().strip().splitlines()[-1]);print($b, d['ms_per_step'],d['value'],d['roofline']['all_mfma_tflops'], d['config']['engine_flags'])"; done
