#!/bin/bash
# A/B of two builds of the library on one box: scripts/ab_libs.sh A.so B.so [bench flags]
# (alternates the two, three rounds each; prints value and ms_per_step).  The builds are selected through APN_LIB_PATH
# (adaptpoint_amd/_lib.py): the shipped libadaptpoint_amd.so is never overwritten, so an interrupted run leaves nothing behind.
# AB_CMD: another command that prints one JSON line with ms_per_step (default: bench.py's headline),
#   e.g. AB_CMD="scripts/bench_pointnext.py --fused --graph" or AB_CMD="bench.py --workload gan".
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=$1; B=$2; shift 2
CMD=${AB_CMD:-"bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary"}
for i in 1 2 3; do
  for x in "$A" "$B"; do
    echo -n "$x  "
    APN_LIB_PATH="$R/$x" APN_ALLOW_UNSAFE_LIB=1 python $R/$CMD "$@" 2>/dev/null \
      | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value', d.get('clouds_per_s')), d['ms_per_step'])"
  done
done
