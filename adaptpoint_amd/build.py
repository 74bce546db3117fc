"""Build recipe for libadaptpoint_amd.so (hipcc, gfx950 only, in-tree).

    python -m adaptpoint_amd.build [--force] [--asm]

Every translation unit is compiled with -ffp-contract=off: the kernels pin
their float rounding with explicit fma builtins (csrc/apn_common.h) -- and
with -fno-slp-vectorize: see the note at CXXFLAGS below (a correctness flag,
not a tuning one).
"""
import argparse
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libadaptpoint_amd.so")
ARCH = "gfx950"
SOURCES = ["capi.hip", "fps.hip", "ball_query.hip", "group_points.hip", "interpolate.hip", "sa_fused.hip", "sa_glue.hip", "sa_seq.hip", "sa_geo.hip", "sa_wide.hip", "sa_wide_glue.hip", "sa_wide_dense.hip", "pointwise.hip", "spectral.hip", "augment.hip", "pointset_group.hip", "attention.hip"]
HEADERS = ["apn_common.h", "apn_mfma.h", "sa_chain.h", "ball_query_body.h",
           os.path.join("..", "..", "include", "adaptpoint_amd.h")]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
            "-fvisibility=hidden", "-Wall", "-Wno-unused-command-line-argument"] + os.environ.get("APN_EXTRA_CXXFLAGS", "").split()
# flags the correctness of the library rests on: a later flag that switches one of them back is refused here, and the
# library carries its flag line (apn_build_flags) so that adaptpoint_amd/_lib.py can refuse a build made without them
REQUIRED_FLAGS = ("-ffp-contract=off", "-fno-slp-vectorize")
FORBIDDEN_FLAGS = ("-fslp-vectorize", "-ffp-contract=fast", "-ffp-contract=on", "-ffast-math", "-Ofast")


# -fno-slp-vectorize, for every translation unit: the SLP vectoriser makes packed-FP32 instructions with operand selection,
# and ONE such form -- v_pk_{add,mul,fma}_f32 with an `op_sel` bit on a VGPR-pair source (the low lane reads the pair's high
# register) -- computes that lane as if the operand were 0.0, now and then, while another stream's MFMA kernels are resident
# (never alone).  Pinned in round 4 by editing the failing builds instruction by instruction and by a synthetic probe:
# profiles/r04_packed_fp32_op_sel.md, DESIGN.md section 4c.  With the form present: wrong FPS picks for ~2 % of the clouds
# beside the benches' MLP stream (rounds 1-3), width-generic features off by 0.1-0.3 beside the fused training step.  The flag
# costs nothing measurable (headline 278.6 k against 279.3 k clouds/s).  csrc/pointwise.hip's explicitly two-wide arithmetic
# (vector types, no operand selection: the form that never failed) stays; tests/test_host_cpu.py disassembles every unit.
NO_SLP = ["-fno-slp-vectorize"]
FILE_FLAGS = {}


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libadaptpoint_amd.so cannot be built")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def flag_line(src):
    return " ".join([*CXXFLAGS, *FILE_FLAGS.get(os.path.basename(src), [])])


def check_flags():
    bad = [f for f in CXXFLAGS if f in FORBIDDEN_FLAGS] + [f for f in REQUIRED_FLAGS if f not in CXXFLAGS]
    if bad:
        raise RuntimeError(f"adaptpoint_amd.build: flag set refused ({bad}): the library's results depend on "
                           f"{REQUIRED_FLAGS} (DESIGN.md section 4c); diagnostic builds go through scripts/asm_variants.py")


def _same_flags(obj, src):
    """An object is stale when the flags it was compiled with differ from today's (recorded next to it)."""
    try:
        return open(obj + ".flags").read() == flag_line(src)
    except OSError:
        return False


def build(force=False, asm=False, verbose=False):
    check_flags()
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in srcs:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and _newer(obj, [src] + hdrs + [os.path.abspath(__file__)]) and _same_flags(obj, src):
            continue
        if os.path.exists(obj + ".flags"):
            os.remove(obj + ".flags")
        stamp = ["-DAPN_BUILD_FLAGS=\"" + flag_line(src) + "\""] if os.path.basename(src) == "capi.hip" else []
        cmd = [hipcc(), f"--offload-arch={ARCH}", *CXXFLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *stamp, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))
        if asm:
            s_out = os.path.join(OBJ, os.path.basename(src)[:-4] + ".s")
            subprocess.run([hipcc(), f"--offload-arch={ARCH}", *CXXFLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "--cuda-device-only",
                            "-S", src, "-o", s_out], check=True)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
        with open(os.path.join(OBJ, os.path.basename(src)[:-4] + ".o.flags"), "w") as fh:
            fh.write(flag_line(src))
    if force or procs or not _newer(LIB, objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--asm", action="store_true", help="also emit device assembly next to the objects")
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.asm, a.verbose))
    sys.exit(0)
