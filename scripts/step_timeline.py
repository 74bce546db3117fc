"""Print the kernel timeline of one steady-state bench step from a rocprofv3 kernel trace.
    python scripts/step_timeline.py gpurun_out/profN/<host>/<pid>_kernel_trace.csv [step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'fps_' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 10
a, b = idx[k], idx[k + 1]
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"dur={(e - s) / 1e3:7.1f}us  {r['Kernel_Name'][:72]}")
    tot += e - s
print(f"sum of kernel durations {tot / 1e3:.1f} us over {b - a} launches; "
      f"span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")
