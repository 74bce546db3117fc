"""Per-kernel duration and the idle gap BEFORE each kernel on its queue, averaged over the steady state of a
rocprofv3 kernel trace (csv): python scripts/chain_gaps.py TRACE.csv FIRST_KERNEL_SUBSTRING [STEPS]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):            # bench.py's per-kernel event pass starts at the first spin kernel: not the steady state
    if "spin_kernel" in r["Kernel_Name"]:
        rows = rows[:i]
        break
first = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
def short(n):
    m = re.search(r"apn::(\w+)", n) or re.search(r"_ZN3apn\d+([A-Za-z_0-9]+?)(?:ILi|E)", n)
    return m.group(1) if m else n[:40]
# the feature stream = everything but the index stage's kernels
SIDE = ("fps_", "ball_query", "tilemap_", "csr_", "sample", "sa_geo")
main = [r for r in rows if not any(k in r["Kernel_Name"] for k in SIDE)]
marks = [i for i, r in enumerate(main) if first in r["Kernel_Name"]]
lo, hi = marks[-steps - 1], marks[-1]
seq = main[lo:hi]
per = (hi - lo) // steps
agg = defaultdict(lambda: [0, 0, 0])
for i, r in enumerate(seq):
    pos = i % per
    gap = int(r["Start_Timestamp"]) - int(seq[i - 1]["End_Timestamp"]) if i else 0
    a = agg[(pos, short(r["Kernel_Name"]))]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[2] += gap
tot_d = tot_g = 0
for (pos, name), (n, d, g) in sorted(agg.items()):
    print(f"{pos:3d} {name:28s} dur {d / n / 1e3:7.1f} us   gap before {g / n / 1e3:6.1f} us")
    tot_d += d / n; tot_g += g / n
print(f"per step: kernels {tot_d / 1e3:.1f} us + gaps {tot_g / 1e3:.1f} us = {(tot_d + tot_g) / 1e3:.1f} us  ({per} launches)")
