set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" = test ]; then
  timeout -k 10 600 python -m pytest $R/tests/test_gpu_fused.py -q -m gpu -x > $O/t.log 2>&1; tail -5 $O/t.log
fi
timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<PY
import json; d=json.load(open("$O/bench_default.json")); print(d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in d["roofline"]["kernels"].items()}); print(d["roofline"]["kernel"], d["roofline"]["frac"])
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
python $R/scripts/steady_stats.py $O/prof/b_kernel_trace.csv sa_prep_stats 20 3 > $O/steady.txt; cat $O/steady.txt
python $R/scripts/chain_gaps.py $O/prof/b_kernel_trace.csv sa_prep_stats 40 > $O/chain.txt; tail -12 $O/chain.txt
rm -rf $O/prof
