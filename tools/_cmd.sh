O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "forward_decisions or tiny_model_logits" > $O/tiny.log 2>&1; echo "tiny rc=$?"; grep -E "tiny|passed|failed|Error" $O/tiny.log | cut -c1-900
timeout -k 10 900 python -m pytest tests/test_full_shape_gpu.py -m gpu -x -q -s -k "across_paths" > $O/full.log 2>&1; echo "full rc=$?"; grep -E "split-bf16 vs float32|passed|failed|Error|assert" $O/full.log | cut -c1-1200
