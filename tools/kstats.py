"""Print a rocprofv3 kernel_stats.csv as per-step microseconds.  usage: kstats.py <csv> <steps> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3
for r in rows[:n]:
    print(f'{float(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us/step {float(r["AverageNs"]) / 1e3:9.1f} avg {r["Calls"]:>6}  {r["Name"][:100]}')
print(f"total {tot / steps:.1f} us/step")
