"""From a rocprofv3 kernel_trace.csv of bench.py: mean duration of each MLP-stream kernel while a sampler kernel
(FPS / ball query) of the index stream is running, and while none is."""
import csv, sys, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
side = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "fps_" in r["Kernel_Name"] or "ball_query" in r["Kernel_Name"])
starts = [s for s, _ in side]
def overlapped(s, e):
    i = bisect.bisect_right(starts, e) - 1
    while i >= 0 and side[i][1] > s - 2_000_000:
        if side[i][0] < e and side[i][1] > s:
            return True
        i -= 1
    return False
acc = collections.defaultdict(lambda: [[0, 0], [0, 0]])
for r in rows:
    n = r["Kernel_Name"]
    if "fps_" in n or "ball_query" in n or "tilemap" in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = acc[n.split("(")[0][-40:]][1 if overlapped(s, e) else 0]
    a[0] += e - s; a[1] += 1
tot = [0.0, 0.0]
for n, (free, busy) in sorted(acc.items(), key=lambda kv: -(kv[1][0][0] + kv[1][1][0])):
    if free[1] and busy[1]:
        f, b = free[0] / free[1] / 1e3, busy[0] / busy[1] / 1e3
        print(f"{n:42s} alone {f:7.1f} us ({free[1]:5d})   beside the sampler {b:7.1f} us ({busy[1]:5d})  x{b / f:.2f}")
