O=gpurun_out/r05o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py -m gpu -x -q -k "wgrad or hwgrad or conv" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 400 python tools/bench_gemm.py --only s1conv --libs htr-vt_amd/lib/libhtrvt_base.so htr-vt_amd/lib/libhtrvt_hip.so --rounds 3 > $O/s1.log 2>&1; echo "rc=$?"; grep wgrad $O/s1.log
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 1 0; do
HTRVT_NO_XCD_RANGES=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('no_xcd_ranges=$v b128',d['ms_per_step'])"
done; done
