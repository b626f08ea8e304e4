O=gpurun_out/r05e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "col_stride" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for s in 5 8 10 16 20 21 24 32 40; do echo "split_h $s"; HTRVT_BENCH_SPLITH=$s timeout -k 10 100 python tools/bench_gemm.py --only sconv --tiles 0 --rounds 2 2>&1 | grep wgrad; done
