"""Summarise rocprofv3 --pmc output per kernel.

    python scripts/pmc_summary.py OUT.csv DIR [DIR ...]

Each DIR holds one `rocprofv3 --pmc … --kernel-trace` pass (`*_counter_collection.csv`).
Writes one row per kernel: dispatches, average duration, and the per-launch average of every
counter found in any pass.  Kernel names are shortened to the function name.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"apn::(\w+)", name)
    if m:
        return m.group(1)
    m = re.search(r"_ZN3apn\d+([A-Za-z_0-9]+?)(?:ILi|E)", name)
    return m.group(1) if m else name[:48]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    key = (path, row["Dispatch_Id"])
                    if key not in seen:
                        seen.add(key)
                        dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    counters = sorted({c for k in vals for c in vals[k]})
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches_per_pass", "avg_us_under_pmc"] + counters)
        for k in sorted(vals):
            n = max(len(v) for v in vals[k].values())
            row = [k, n, round(sum(dur[k]) / max(len(dur[k]), 1), 2)]
            for c in counters:
                v = vals[k].get(c)
                row.append(round(sum(v) / len(v), 1) if v else "")
            w.writerow(row)
    print(open(out).read())


if __name__ == "__main__":
    main()
