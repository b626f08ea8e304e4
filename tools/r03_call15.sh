#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_stem_gpu.py -m gpu -x -q > $O/t15.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/t15.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path > $O/b128_15.json 2> $O/b128_15.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_15.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof15 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-path --no-overlap-wgrad > $GRAFT_REPO_ROOT/$O/prof15.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py $O/prof15 $O/prof15.md --steps 7 > /dev/null 2>&1; grep -E "stem|conv1_bwd|relayout|adamw|Sum of" $O/prof15.md | head -20
