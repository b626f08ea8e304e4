O=gpurun_out/r05s; mkdir -p $O
L=htr-vt_amd/lib
timeout -k 10 400 python tools/bench_gemm.py --only enc --libs $L/libhtrvt_base.so $L/libhtrvt_st3.so $L/libhtrvt_st5.so --rounds 3 > $O/enc.log 2>&1; echo "rc=$?"; grep -E "gelu" $O/enc.log
