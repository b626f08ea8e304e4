O=gpurun_out/r05k; mkdir -p $O
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in 100000 300 150 0; do
HTRVT_BN_STREAM_MB=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-path > $O/ab_${v}_$rep.json 2>$O/ab.err; python -c "import json;d=json.load(open('$O/ab_${v}_$rep.json'));print('stream_mb=$v b128',d['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
for v in 100000 150; do
export HTRVT_BN_STREAM_MB=$v
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity-path --no-overlap-wgrad > $R/$O/prof_$v.log 2>&1; echo "prof $v rc=$?"
done
