"""Print a rocprofv3 kernel_stats.csv as per-step microseconds.  usage: kstats.py <csv> <steps|auto> [rows]
auto: the number of steps the trace holds = the call count of the optimizer kernel (exactly one launch per step, eager or
replayed) -- round 5's hard-coded 46 divided a 95-step trace (the us/step column summed to 3x the step)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if sys.argv[2] == "auto":
    once = [r for r in rows if "sg_rmsprop_kernel" in r["Name"] or "sg_adam_kernel" in r["Name"]]
    steps = float(sum(int(r["Calls"]) for r in once)) if once else 1.0
else:
    steps = float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3
print(f"# {steps:.0f} steps in the trace; us/step = total duration / steps (sums GPU time over both branches of the step: it "
      "exceeds the step's wall time where launches overlap)")
for r in rows[:n]:
    print(f'{float(r["TotalDurationNs"]) / 1e3 / steps:9.1f} us/step {float(r["AverageNs"]) / 1e3:9.1f} avg '
          f'{int(r["Calls"]) / steps:6.2f} calls/step  {r["Name"][:100]}')
print(f"total {tot / steps:.1f} us/step")
