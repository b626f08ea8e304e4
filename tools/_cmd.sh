mkdir -p gpurun_out/r05f
timeout -k 10 300 python -m pytest tests/test_gemm8p_gpu.py -m gpu -x -q -k "any_split_factor" > gpurun_out/r05f/split_test.log 2>&1; echo "split test rc=$?"; tail -3 gpurun_out/r05f/split_test.log
bash tools/round_evidence.sh r05f bench
