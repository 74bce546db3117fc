#!/bin/bash
# Runs the beside-another-stream bit-exactness tests ONCE per diagnostic build of scripts/asm_variants.py and writes a
# table.  Expected failures are test failures, not hangs: a timeout ends the whole run.
#   scripts/run_asm_variants.sh OUTFILE unit[:test-expression] ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
: > "$OUT"
for spec in "$@"; do
  unit=${spec%%:*}; expr=${spec#*:}
  for v in ${APN_VARIANTS:-shipping slp sel2scalar all2scalar nop}; do
    if [ $v = shipping ]; then unset APN_LIB_PATH APN_ALLOW_UNSAFE_LIB
    else export APN_LIB_PATH=$R/adaptpoint_amd/variants/libadaptpoint_amd_${unit}_$v.so APN_ALLOW_UNSAFE_LIB=1; fi
    log=$(mktemp)
    timeout -k 10 300 python -m pytest "$R/tests/test_gpu_concurrency.py" -q -x -k "$expr" -p no:cacheprovider > "$log" 2>&1
    rc=$?
    echo "unit=$unit variant=$v rc=$rc  $(grep -E 'passed|failed' "$log" | tail -1)" | tee -a "$OUT"
    grep -E "^E .*(differ|deviation|round, tensor)" "$log" | head -3 | cut -c1-400 >> "$OUT"
    rm -f "$log"
    if [ $rc -ge 124 ]; then echo "timeout: stopping" | tee -a "$OUT"; exit 1; fi
  done
done
exit 0
