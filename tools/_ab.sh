set -e
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"
run() { python bench.py --no-parity-path --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "two kernels  $(run --no-fuse-stem)"
  echo "fused stem   $(run)"
done
echo "fwd two kernels $(run --forward-only --no-fuse-stem)"
echo "fwd fused       $(run --forward-only)"
