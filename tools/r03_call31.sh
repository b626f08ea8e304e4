#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t31.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/t31.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path > $O/b128_31.json 2> $O/b128_31.err; python -c "
import json;d=json.loads(open('$O/b128_31.json').read().strip().splitlines()[-1]);print(128, d['ms_per_step'],d['value'])"
