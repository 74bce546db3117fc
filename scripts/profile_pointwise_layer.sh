#!/bin/bash
# Per-kernel durations of one per-point layer (fwd + bwd) under rocprof: gpurun_out/<tag>/pw_<layer>_kernels.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${APN_ROUND_TAG:-r03}
L=${1:-decode1}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof_pw -o p -- python $R/scripts/bench_pointwise.py --layers $L --only planes3 --iters 3 > $O/prof_pw.log 2>&1
python - "$O/prof_pw/p_kernel_trace.csv" > $O/pw_${L}_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows[len(rows) // 2:]:
    name = r["Kernel_Name"].split("(")[0][:60]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    print("%-62s grid %6s %4s %4s wg %4s  n=%4d  avg %8.1f us" % (k + (len(v), sum(v) / len(v))))
PY
rm -rf $O/prof_pw
cat $O/pw_${L}_kernels.txt
