"""Per-step kernel statistics of the STEADY STATE from a rocprofv3 kernel trace (csv).

    python scripts/steady_stats.py TRACE.csv MARKER PER_STEP [STEPS] [--csv OUT.csv]

A library's first calls per shape (MIOpen's solver search, Tensile warm-up) pollute whole-run
statistics.  The steps are delimited by a MARKER kernel (substring of its name) that runs
PER_STEP times per step; the last STEPS (default 3) whole steps are summarised: per kernel name
calls/step, average duration and time/step, plus the step's kernel-time sum, launch count and span.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"apn::(\w+)", name) or re.search(r"_ZN3apn\d+([A-Za-z_0-9]+?)(?:ILi|E)", name)
    if m:
        return "apn::" + m.group(1)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_csv = None
    if "--csv" in sys.argv:
        out_csv = sys.argv[sys.argv.index("--csv") + 1]
        args = [a for a in args if a != out_csv]
    trace, marker, per_step = args[0], args[1], int(args[2])
    steps = int(args[3]) if len(args) > 3 else 3
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # bench.py's per-kernel event pass (after the timed region) enqueues every eager step behind a spin kernel: the
    # replayed steady state ends at the first of them
    for i, r in enumerate(rows):
        if "spin_kernel" in r["Kernel_Name"]:
            rows = rows[:i]
            break
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    bounds = marks[::per_step]
    lo, hi = bounds[-steps - 1], bounds[-1]
    sel = rows[lo:hi]
    agg = defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / steps / 1e6
    busy = sum(v[1] for v in agg.values()) / steps / 1e6
    print(f"steady state over the last {steps} steps: {len(sel) / steps:.0f} launches/step, "
          f"kernel-time sum {busy:.3f} ms/step, wall span {span:.3f} ms/step")
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for name, (calls, ns) in table[:40]:
        print(f"{ns / steps / 1e6:8.3f} ms/step {calls / steps:7.1f} calls/step avg {ns / calls / 1e3:8.1f} us  {name}")
    if out_csv:
        with open(out_csv, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "CallsPerStep", "AverageNs", "NsPerStep"])
            for name, (calls, ns) in table:
                w.writerow([name, round(calls / steps, 2), round(ns / calls, 1), round(ns / steps, 1)])
            w.writerow(["TOTAL", round(len(sel) / steps, 1), "", round(busy * 1e6, 1)])
            w.writerow(["WALL_SPAN_PER_STEP", "", "", round(span * 1e6, 1)])


if __name__ == "__main__":
    main()
