"""Diagnostic builds for DESIGN.md section 4c (the packed-FP32 question): the SAME vectorised device assembly of one
translation unit, edited instruction by instruction, assembled and linked with the shipping objects of all others.

    python scripts/asm_variants.py [--unit fps]       # -> adaptpoint_amd/variants/libadaptpoint_amd_<variant>.so

Variants of the unit's device code (register allocation and schedule are those of the vectorised build in all of them):
  slp        the compiler's vectorised assembly, unedited                      (control: the build known to fail)
  sel2scalar every v_pk_{add,mul,fma}_f32 that carries an op_sel / op_sel_hi modifier -> two scalar VALU instructions
  all2scalar every v_pk_{add,mul,fma}_f32 -> two scalar VALU instructions
  nop        unedited instructions, `s_nop 1` in front of every v_pk_*_f32      (two extra wait states, nothing else)
second pass (--narrow), to say WHICH operand-selected form matters:
  selhi2scalar  only the forms with op_sel_hi and no op_sel (a low half broadcast to both lanes) -> scalar
  sello2scalar  only the forms with op_sel (a high half feeding the low lane) -> scalar
  vsel2scalar   only the forms whose non-default selection applies to a VGPR pair -> scalar (SGPR-sourced ones kept)
  ssel2scalar   only the forms whose non-default selection applies to an SGPR pair -> scalar (VGPR-sourced ones kept)
third pass (--overlap), after the second blamed the op_sel forms on VGPR pairs:
  ovl2scalar    only the op_sel forms whose DESTINATION pair is the pair the op_sel applies to -> scalar
  novl2scalar   only the op_sel forms whose destination is another pair -> scalar
Select one at run time with APN_LIB_PATH=<path> APN_ALLOW_UNSAFE_LIB=1 (adaptpoint_amd/_lib.py).
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adaptpoint_amd import build as B  # noqa: E402

LLVM = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(ROOT, "adaptpoint_amd", "variants")
PK = re.compile(r"^(\s*)v_pk_(add|mul|fma)_f32\s+(.*)$")
MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def split_operands(text):
    mods = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in MOD.finditer(text)}
    ops = [o.strip() for o in MOD.sub("", text).split(",") if o.strip()]
    return ops, mods


def half(op, hi):
    """The 32-bit half of a 64-bit operand: v[a:b] / s[a:b] -> v<a+hi> / s<a+hi>; constants apply to both halves."""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + hi}"
    return op


def scalarise(kind, ops, mods):
    """Two scalar instructions computing what the packed one computes; None when the destination overlaps the sources in
    a way no ordering of the two resolves."""
    nsrc = 3 if kind == "fma" else 2
    dst, srcs = ops[0], ops[1:1 + nsrc]
    sel = mods.get("op_sel", [0] * nsrc)
    sel_hi = mods.get("op_sel_hi", [1] * nsrc)
    neg_lo = mods.get("neg_lo", [0] * nsrc)
    neg_hi = mods.get("neg_hi", [0] * nsrc)
    name = {"add": "v_add_f32_e64", "mul": "v_mul_f32_e64", "fma": "v_fma_f32"}[kind]

    def one(dhi, sels, negs):
        a = [("-" if n else "") + half(s, h) for s, h, n in zip(srcs, sels, negs)]
        reads = {half(s, h) for s, h in zip(srcs, sels)}
        return half(dst, dhi), f"{name} {half(dst, dhi)}, " + ", ".join(a), reads
    lo = one(0, sel, neg_lo)
    hi = one(1, sel_hi, neg_hi)
    if lo[0] not in hi[2]:
        return [lo[1], hi[1]]
    if hi[0] not in lo[2]:
        return [hi[1], lo[1]]
    return None


def variant(lines, which):
    out, n_edit, n_kept = [], 0, 0
    for ln in lines:
        m = PK.match(ln)
        if not m:
            out.append(ln)
            continue
        indent, kind, rest = m.groups()
        rest = rest.split(";")[0].rstrip()
        ops, mods = split_operands(rest)
        has_sel = "op_sel" in mods or "op_sel_hi" in mods
        nsrc = 3 if kind == "fma" else 2
        picked = [i for i in range(nsrc) if mods.get("op_sel", [0] * nsrc)[i] != 0 or mods.get("op_sel_hi", [1] * nsrc)[i] != 1]
        on_vgpr = any(ops[1 + i].startswith("v[") for i in picked)
        on_sgpr = any(ops[1 + i].startswith("s[") for i in picked)
        dst_is_picked = any(ops[1 + i] == ops[0] for i in range(nsrc) if mods.get("op_sel", [0] * nsrc)[i] != 0)
        narrow = {"ovl2scalar": "op_sel" in mods and dst_is_picked, "novl2scalar": "op_sel" in mods and not dst_is_picked,
                  "selhi2scalar": "op_sel_hi" in mods and "op_sel" not in mods, "sello2scalar": "op_sel" in mods,
                  "vsel2scalar": has_sel and on_vgpr, "ssel2scalar": has_sel and on_sgpr and not on_vgpr}
        if which == "nop":
            out.append(f"{indent}s_nop 1\n")
            out.append(ln)
            n_edit += 1
            continue
        if which == "all2scalar" or (which == "sel2scalar" and has_sel) or narrow.get(which, False):
            two = scalarise(kind, ops, mods)
            if two is None:
                raise RuntimeError("cyclic overlap: " + ln)
            out += [f"{indent}{t}\n" for t in two]
            n_edit += 1
        else:
            out.append(ln)
            n_kept += 1
    return out, n_edit, n_kept


def run(cmd):
    subprocess.run(cmd, check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="fps", help="comma-separated translation units (fps,sa_wide,...)")
    ap.add_argument("--narrow", action="store_true", help="the second pass's four variants instead of the first pass's")
    ap.add_argument("--overlap", action="store_true", help="the third pass's two variants")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    B.build()                                                      # the shipping objects of every other unit
    units = a.unit.split(",")
    for a.unit in units:
        one_unit(a)


def one_unit(a):
    src = os.path.join(B.CSRC, a.unit + ".hip")
    flags = [f for f in B.CXXFLAGS if f != "-fno-slp-vectorize"]
    base = os.path.join(OUT, a.unit + "_slp.s")
    run([B.hipcc(), f"--offload-arch={B.ARCH}", *flags, "--cuda-device-only", "-S", src, "-o", base])
    lines = open(base).readlines()
    others = [os.path.join(B.OBJ, s[:-4] + ".o") for s in B.SOURCES if s != a.unit + ".hip"]
    for which in (("ovl2scalar", "novl2scalar") if a.overlap else ("selhi2scalar", "sello2scalar", "vsel2scalar", "ssel2scalar") if a.narrow
                  else ("slp", "sel2scalar", "all2scalar", "nop")):
        text, n_edit, n_kept = (lines, 0, sum(1 for ln in lines if PK.match(ln))) if which == "slp" else variant(lines, which)
        stem = os.path.join(OUT, f"{a.unit}_{which}")
        with open(stem + ".s", "w") as fh:
            fh.writelines(text)
        run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={B.ARCH}", "-c", stem + ".s", "-o", stem + ".dev.o"])
        run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", stem + ".out", stem + ".dev.o"])
        run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
             f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{B.ARCH}", "-input=/dev/null",
             "-input=" + stem + ".out", "-output=" + stem + ".hipfb"])
        run([B.hipcc(), f"--offload-arch={B.ARCH}", *flags, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary",
             "-Xclang", stem + ".hipfb", "-c", src, "-o", stem + ".o"])
        lib = os.path.join(OUT, f"libadaptpoint_amd_{a.unit}_{which}.so")
        run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", stem + ".o", *others, "-o", lib])
        print(f"{which:11s} packed-FP32 instructions edited {n_edit:5d}, kept {n_kept:5d} -> {os.path.relpath(lib, ROOT)}")
        for ext in (".dev.o", ".out", ".hipfb", ".o"):
            os.remove(stem + ext)


if __name__ == "__main__":
    main()
