#!/bin/bash
# per-kernel times of the per-point MLP layer (scripts/bench_pointwise.py) for the layers given as arguments
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pw_$L -o pw -- python3 $R/scripts/bench_pointwise.py --layers $L --only ${ONLY:-planes3} --iters 5 > $O/prof_pw_$L.log 2>&1 || exit 1
    echo "== $L"
    python3 - <<PY
import csv, glob
f = glob.glob("$O/prof_pw_$L/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f'{r["Name"][:80]:80s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
    rm -rf $O/prof_pw_$L
done
