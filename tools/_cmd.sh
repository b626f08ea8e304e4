O=gpurun_out/r05n; mkdir -p $O
timeout -k 10 300 python tools/bench_gemm.py --only c1x1 --tiles 0 8 7 3 --rounds 2 > $O/c1.log 2>&1; echo "rc=$?"; grep wgrad $O/c1.log
for s in 64 128 256 512; do echo "split $s"; HTRVT_BENCH_SPLITK=$s timeout -k 10 100 python tools/bench_gemm.py --only c1x1 --tiles 8 3 --rounds 2 2>&1 | grep wgrad; done
