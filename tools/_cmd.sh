O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_aux_gpu.py -m gpu -x -q -k "colsum" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_train_iter_gpu.py tests/test_dp_gpu.py tests/test_graph_gpu.py -m gpu -x -q > $O/tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 $O/tests2.log
for rep in 1 2; do
for m in 0 1; do HTRVT_COLSUM_TWO_LAUNCHES=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_$m_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_$m_$rep.json'));print('two_launches=$m b128',d['ms_per_step'])"
HTRVT_COLSUM_TWO_LAUNCHES=$m timeout -k 10 200 python bench.py --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab16_$m_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab16_$m_$rep.json'));print('two_launches=$m b16',d['ms_per_step'])"
done; done
