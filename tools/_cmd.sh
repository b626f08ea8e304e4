mkdir -p gpurun_out/r05v
O=gpurun_out/r05v
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py tests/test_stem_gpu.py -m gpu -x -q -k "conv or halo or stem" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 300 python tools/stamp_probe_halo.py htr-vt_amd/lib/libhtrvt_stamp.so 2>&1 | grep -A8 "forward" | grep -E "forward|walk|tile total"
timeout -k 10 400 python tools/bench_gemm.py --only "s1conv sconv c1x1" --libs htr-vt_amd/lib/libhtrvt_base.so htr-vt_amd/lib/libhtrvt_hip.so --rounds 3 2>&1 | grep fwd
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in base hip; do
HTRVT_LIB=$R/htr-vt_amd/lib/libhtrvt_$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('$v b128',d['ms_per_step'])"
done; done
