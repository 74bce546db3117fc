"""Per-call durations (and grids) of the kernels whose name contains argv[2] in a rocprofv3 kernel_trace.csv,
last argv[3] calls in time order."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-int(sys.argv[3]):]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")) if k in r)
    print(f"{d:8.1f} us  grid {grid:12s} lds {r.get('LDS_Block_Size', '?'):>7s}  {r['Kernel_Name'][:60]}")
