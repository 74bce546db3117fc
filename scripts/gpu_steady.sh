#!/bin/bash
# Per-kernel durations INSIDE the replayed step (rocprofv3 kernel trace + scripts/steady_stats.py), short form of
# `collect_profiles.sh bench`:   bash scripts/gpu_steady.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${APN_ROUND_TAG:-r05}
mkdir -p $O
tag=${1:-s}
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o b -- python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary "$@" > $O/${tag}_prof.log 2>&1 || { tail -20 $O/${tag}_prof.log; exit 1; }
python $R/scripts/steady_stats.py $O/prof_$tag/b_kernel_trace.csv sa_prep_stats 20 3 > $O/${tag}_steady.txt
cp $O/prof_$tag/b_kernel_stats.csv $O/${tag}_kernel_stats.csv
rm -rf $O/prof_$tag
grep '^{' $O/${tag}_prof.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.1f ms %.4f' % (d['value'], d['ms_per_step']))"
head -12 $O/${tag}_steady.txt
