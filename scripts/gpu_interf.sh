#!/bin/bash
# bench under a kernel trace + scripts/index_interference.py:  bash scripts/gpu_interf.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${APN_ROUND_TAG:-r05}
mkdir -p $O
tag=${1:-i}
shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o b -- python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary "$@" > $O/${tag}_prof.log 2>&1 || { tail -20 $O/${tag}_prof.log; exit 1; }
grep '^{' $O/${tag}_prof.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.1f ms %.4f' % (d['value'], d['ms_per_step']))"
python $R/scripts/index_interference.py /tmp/prof_$tag/b_kernel_trace.csv | tee $O/${tag}_index_interference.txt
python $R/scripts/steady_stats.py /tmp/prof_$tag/b_kernel_trace.csv sa_prep_stats 20 3 > $O/${tag}_steady.txt
cp /tmp/prof_$tag/b_kernel_stats.csv $O/${tag}_kernel_stats.csv
