O=gpurun_out/r05h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "halo or fused" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 400 python tools/bench_gemm.py --only s1conv --tiles 21 20 --rounds 3 > $O/persist_ab.log 2>&1; echo "ab rc=$?"; cat $O/persist_ab.log
timeout -k 10 400 python tools/bench_gemm.py --only fdgrad --tiles 21 20 --rounds 2 > $O/persist_fd.log 2>&1; echo "fd rc=$?"; cat $O/persist_fd.log
