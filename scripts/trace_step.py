"""Print the kernel timeline of one mid-run step from a rocprofv3 kernel trace csv.

    python scripts/trace_step.py gpurun_out/profX/run_kernel_trace.csv [step_index]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
idx = [i for i, r in enumerate(rows) if "fps_" in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 3:i1 + 1]:
    n = r["Kernel_Name"]
    m = re.search(r"apn::(\w+)|_ZN3apn\d+(\w+?)E", n)
    nm = (m.group(1) or m.group(2)) if m else n[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{r.get('Queue_Id', '?')} {nm}")
