"""Timeline of the LAST n launches of a rocprofv3 kernel trace (one replay of a captured step holds a known
number of kernel nodes): start offset, duration, the gap to the end of everything launched before, name.
    python scripts/tail_timeline.py TRACE.csv N_LAUNCHES > out.txt"""
import csv
import re
import sys


def short(name):
    m = re.search(r"apn::(\w+)", name)
    if m:
        return "apn::" + m.group(1)
    return re.sub(r"^void ", "", name)[:100]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
if n <= 0:
    # the trace ends with replays of one graph: the period is the distance between the last launch of MARKER (argv[3],
    # default fps_atomic) and the one PER_STEP (argv[4], default 7) launches before it; one period is then cut
    # starting behind the largest idle gap inside the last two periods
    marker = sys.argv[3] if len(sys.argv) > 3 else "fps_atomic"
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    hits = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
    best = hits[-1] - hits[-1 - per]
    skip = 8                                    # the run's last launches (result copies) are not part of a replay
    n = best
    lo = len(rows) - skip - 2 * n
    gaps = [(int(rows[i]['Start_Timestamp']) - max(int(r['End_Timestamp']) for r in rows[max(lo - 50, 0):i]), i)
            for i in range(lo, lo + n)]
    start = max(gaps)[1]
    rows = rows[start:start + n]
else:
    rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
end = t0
busy = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = s - end
    if e > end:
        busy += e - max(s, end)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {short(r['Kernel_Name'])}")
    end = max(end, e)
print(f"# span {(end - t0) / 1e3:.1f} us, device busy (union of kernels) {busy / 1e3:.1f} us, {n} launches")
