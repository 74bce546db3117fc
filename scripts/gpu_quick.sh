#!/bin/bash
# One short GPU-box pass for kernel iterations: a subset of the GPU tests, then the headline bench with its per-kernel
# table.   bash scripts/gpu_quick.sh <tag> ["pytest args" | -] [bench args]
# Writes gpurun_out/r05/<tag>_{tests.log,bench.json,bench.err}; prints value / ms_per_step / per-kernel microseconds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${APN_ROUND_TAG:-r05}
mkdir -p $O
tag=${1:-q}
tests=${2:--}
shift 2 2>/dev/null
cd $R
if [ "$tests" != "-" ]; then
    python -m pytest $tests -x -q > $O/${tag}_tests.log 2>&1
    rc=$?
    tail -4 $O/${tag}_tests.log
    if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $O/${tag}_tests.log | head -20; exit $rc; fi
fi
python bench.py --no-cpu-baseline --no-secondary "$@" > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { tail -20 $O/${tag}_bench.err; exit 1; }
python - "$O/${tag}_bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.1f  ms_per_step %.4f  frac %.4f  verified idx %s mlp %s" % (
    d["value"], d["ms_per_step"], r["frac"], r["index_stream"]["verified_after_timed_region"],
    r["index_stream"]["mlp_stream_verified_after_timed_region"]))
print({k: v["avg_us"] for k, v in r["kernels"].items()})
PY
