O=gpurun_out/r05l; mkdir -p $O
timeout -k 10 500 python tools/bench_gemm.py --only fdgrad --libs htr-vt_amd/lib/libhtrvt_base.so htr-vt_amd/lib/libhtrvt_hip.so --rounds 3 > $O/fd.log 2>&1; echo "fd rc=$?"; cat $O/fd.log
timeout -k 10 300 python tools/bench_gemm.py --only sdgrad --libs htr-vt_amd/lib/libhtrvt_base.so htr-vt_amd/lib/libhtrvt_hip.so --rounds 2 > $O/sd.log 2>&1; echo "sd rc=$?"; cat $O/sd.log
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in base hip; do
HTRVT_LIB=$R/htr-vt_amd/lib/libhtrvt_$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('$v b128',d['ms_per_step'])"
done; done
