#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_stem_gpu.py tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py -m gpu -x -q > $O/t13.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/t13.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path --gemm-table $O/gemm_table13.txt > $O/b128_13.json 2> $O/b128_13.err; echo "bench rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_13.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'],d['roofline']['kernel'],d['roofline']['achieved'],d['roofline']['mfma_ms_per_step'])"
HTRVT_STEM_VALU=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-path > $O/b128_13v.json 2> $O/b128_13v.err; echo "bench valu-stem rc=$?"; python -c "
import json;d=json.loads(open('$O/b128_13v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['value'])"
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof13 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity-path --no-overlap-wgrad > $GRAFT_REPO_ROOT/$O/prof13.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py $O/prof13 $O/prof13.md --steps 7 > /dev/null 2>&1; grep -E "stem|gemm8p|conv1_bwd|relayout|bn_bwd_apply|adamw" $O/prof13.md | head -20
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t13_all.log 2>&1; echo "all tests rc=$?"; tail -4 $O/t13_all.log
