"""Top rows of a rocprofv3 --stats kernel_stats.csv found under the given directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", round(tot / 1e6, 3))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-90s %6s %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
