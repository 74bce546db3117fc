#!/bin/bash
# Diagnostic builds of csrc/pointwise.hip with parts of the contraction kernel's main loop knocked out (-DPW_KNOCK=mask:
# 1 no MFMAs, 2 no split / LDS writes, 4 no global loads, 8 no fragment reads), linked with the shipping objects of every
# other unit into adaptpoint_amd/variants/ (git-ignored), and the three contractions of one layer timed per build:
#     bash scripts/pw_knockouts.sh build          (here: hipcc cross-compiles)
#     bash scripts/pw_knockouts.sh run            (on the GPU box)
# Results are wrong by construction; only the times mean anything.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/adaptpoint_amd/variants
MASKS=${PW_MASKS:-"0 1 2 4 8 3 6 9 14 15"}
if [ "$1" = build ]; then
  mkdir -p "$V"
  python -m adaptpoint_amd.build >/dev/null || exit 1
  FLAGS=$(cat "$R/adaptpoint_amd/csrc/build/pointwise.o.flags")
  OTHERS=$(ls "$R"/adaptpoint_amd/csrc/build/*.o | grep -v pointwise.o)
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -DPW_KNOCK=$m -c "$R/adaptpoint_amd/csrc/pointwise.hip" -o "$V/pw_k$m.o" || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS "$V/pw_k$m.o" -o "$V/libpw_k$m.so" || exit 1
    rm -f "$V/pw_k$m.o"
  done
  ls -la "$V"
else
  shift
  for m in $MASKS; do
    echo -n "knock=$m  "
    APN_LIB_PATH="$V/libpw_k$m.so" APN_ALLOW_UNSAFE_LIB=1 python "$R/scripts/bench_pw_gemm.py" "$@" 2>/dev/null | tail -1
  done
fi
