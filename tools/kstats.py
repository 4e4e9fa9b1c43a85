"""Print selected rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [substring ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    name = r.get("Name", "")
    if pats and not any(p in name for p in pats):
        continue
    print("%-48s calls %6s  avg %10.2f us  total %10.1f us  %5s %%" % (
        name[:48], r.get("Calls"), float(r.get("AverageNs", 0)) / 1e3, float(r.get("TotalDurationNs", 0)) / 1e3,
        r.get("Percentage", "")[:5]))
