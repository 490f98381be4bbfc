"""Timeline of ONE train step from a rocprofv3 kernel_trace.csv of bench.py: for every dispatch of the last complete
step: start offset, duration, gap to the previous end, stream/queue id.   usage: step_timeline.py <dir>"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the window gather kernel
starts = [i for i, r in enumerate(rows) if "sg_window_gather" in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy_end = t0
idle = 0.0
print(f"{len(step)} dispatches, step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].replace("void ", "")[:58]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  q{q}  {name}")
    busy_end = max(busy_end, e)
print(f"GPU idle inside the step (no kernel running): {idle:.1f} us")
