#!/bin/bash
# Kernel timeline of one replayed joint step: gpurun_out/<tag>/gan_timeline[_overlap].txt
# usage: scripts/profile_gan_timeline.sh NODES [bench flags]     (NODES = launches per replay, 0 = detect the period)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${APN_ROUND_TAG:-r03}
O=$R/gpurun_out/$T
N=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_gt -o b -- python $R/scripts/bench_gan_step.py --mode fused --graph --iters 8 --warmup 2 "$@" > $O/prof_gt.log 2>&1
python $R/scripts/tail_timeline.py $O/prof_gt/b_kernel_trace.csv $N > $O/gan_timeline.txt
cp $O/prof_gt/b_kernel_trace.csv $O/gan_trace.csv; rm -rf $O/prof_gt
tail -2 $O/gan_timeline.txt
