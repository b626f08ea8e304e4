O=gpurun_out/r05t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_iter_gpu.py tests/test_dp_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2 3; do for v in 1 0; do
HTRVT_NO_OPT_OVERLAP=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('no_opt_overlap=$v b128',d['ms_per_step'])"
HTRVT_NO_OPT_OVERLAP=$v timeout -k 10 200 python bench.py --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab16_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab16_${v}_$rep.json'));print('no_opt_overlap=$v b16',d['ms_per_step'])"
done; done
