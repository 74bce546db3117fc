"""From a rocprofv3 kernel_trace.csv of bench.py: what each index-stream kernel costs the MLP-stream kernels that run
beside it.  Every MLP-stream launch is classified by the index-stream kernel that overlaps most of its duration (or
'alone'); printed per MLP kernel: mean duration per class, and per class the extra microseconds per 20-step replay.

    python scripts/index_interference.py <kernel_trace.csv>
"""
import bisect
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
MLP = ("sa_prep_stats", "sa_fwd_main", "fwd_out", "bwd_prep", "sa_bwd_kernel", "bwd_point_grads", "bwd_finalize")


def short(n):
    for k in MLP:
        if k in n:
            return k
    return None


def side_name(n):
    for k in ("fps_", "ball_query", "sa_geo", "tilemap_fill", "tilemap_pack", "tilemap", "csr"):
        if k in n:
            return k.rstrip("_")
    return "other"


side = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), side_name(r["Kernel_Name"]))
              for r in rows if short(r["Kernel_Name"]) is None)
starts = [s for s, _, _ in side]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
t_first = min(int(r["Start_Timestamp"]) for r in rows)
t_last = max(int(r["End_Timestamp"]) for r in rows)
skip_before = t_first + (t_last - t_first) * 0.25          # warm-up / capture
for r in rows:
    k = short(r["Kernel_Name"])
    if k is None:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < skip_before:
        continue
    cover = collections.defaultdict(int)
    i = bisect.bisect_right(starts, e) - 1
    while i >= 0 and side[i][0] > s - 3_000_000:
        a, b, nm = side[i]
        ov = min(e, b) - max(s, a)
        if ov > 0:
            cover[nm] += ov
        i -= 1
    cls = "alone"
    if cover:
        nm, ov = max(cover.items(), key=lambda kv: kv[1])
        if ov > 0.3 * (e - s):
            cls = nm
    a = acc[k][cls]
    a[0] += e - s
    a[1] += 1
classes = sorted({c for k in acc for c in acc[k]}, key=lambda c: (c != "alone", c))
print("%-18s" % "mean us (count)" + "".join("%22s" % c for c in classes))
extra = collections.defaultdict(float)
total = collections.defaultdict(int)
for k in MLP:
    if k not in acc:
        continue
    base = acc[k]["alone"][0] / max(acc[k]["alone"][1], 1) / 1e3
    line = "%-18s" % k
    for c in classes:
        t, n = acc[k][c]
        line += "%22s" % ("%.1f (%d)" % (t / n / 1e3, n) if n else "-")
        if n and c != "alone" and base > 0:
            extra[c] += (t / 1e3 - n * base)
            total[c] += n
    print(line)
n_steps = sum(acc["sa_bwd_kernel"][c][1] for c in classes)
print("extra microseconds per step, by the index kernel beside which they were spent (over %d steps):" % n_steps)
for c in classes:
    if c != "alone":
        print("  %-14s %6.2f us/step   (%d MLP launches beside it)" % (c, extra[c] / max(n_steps, 1), total[c]))
print("  %-14s %6.2f us/step" % ("all", sum(extra.values()) / max(n_steps, 1)))
