#!/bin/bash
# Quick look at the default step's kernel times: gpurun_out/<tag>/quick_steady.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${APN_ROUND_TAG:-r03}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary $@"
python $R/bench.py --steps 2000 --warmup 200 $B 2>/dev/null | grep '^{' | cut -c1-220
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -o b -- python $R/bench.py --steps 2000 --warmup 200 $B > $O/prof_q.log 2>&1
python $R/scripts/steady_stats.py $O/prof_q/b_kernel_trace.csv sa_prep_stats 20 3 > $O/quick_steady.txt
rm -rf $O/prof_q
head -14 $O/quick_steady.txt
