O=gpurun_out/r05m; mkdir -p $O
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in base hip; do
HTRVT_LIB=$R/htr-vt_amd/lib/libhtrvt_$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('$v b128',d['ms_per_step'])"
done; done
