#!/bin/bash
# Kernel times of the headline step with bit-reproducible gradients (bench.py --deterministic) -> gpurun_out/<tag>/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${APN_ROUND_TAG:-r03}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-secondary --deterministic"
python $R/bench.py --steps 2000 --warmup 200 $B 2>/dev/null | grep '^{' > $O/${T}_bench_deterministic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_det -o b -- python $R/bench.py --steps 2000 --warmup 200 $B > $O/prof_det.log 2>&1
cp $O/prof_det/b_kernel_stats.csv $O/${T}_bench_deterministic_kernel_stats.csv
python $R/scripts/steady_stats.py $O/prof_det/b_kernel_trace.csv sa_prep_stats 20 3 > $O/${T}_bench_deterministic_steady.txt
rm -rf $O/prof_det
