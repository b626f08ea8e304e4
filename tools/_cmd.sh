O=gpurun_out/r05r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py tests/test_gemm_gpu.py -m gpu -x -q -k "wgrad or hwgrad or conv or split" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for v in 1 0; do echo "legacy=$v"; HTRVT_SPLITK_LEGACY=$v timeout -k 10 300 python tools/bench_gemm.py --only "s1conv sconv lwgrad_proj" --rounds 2 2>&1 | grep wgrad; done
for rep in 1 2 3; do for v in 1 0; do
HTRVT_SPLITK_LEGACY=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('legacy=$v b128',d['ms_per_step'])"
done; done
