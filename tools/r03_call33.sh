#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
for i in 1 2 3; do for ov in "" "deterministic=0"; do
HTRVT_ENGINE_OVERRIDE=$ov timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/b128_33.json 2> $O/b128_33.err; python -c "
import json;d=json.loads(open('$O/b128_33.json').read().strip().splitlines()[-1]);print('[$ov]', d['ms_per_step'],d['value'],d['config']['engine_flags']['deterministic'])"; done; done
