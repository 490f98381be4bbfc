"""Print per-dispatch durations from a rocprofv3 kernel_trace.csv.  usage: trace_rows.py <dir> <pattern> [rows]"""
import csv
import glob
import sys

pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = [r for r in rows if pat in r["Kernel_Name"]]
for r in out[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{d:8.1f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?"))} wg {r.get("Workgroup_Size_X", "?")}  '
          f'lds {r.get("LDS_Block_Size", "?")} vgpr {r.get("VGPR_Count", "?")} sgpr {r.get("SGPR_Count", "?")}  {r["Kernel_Name"][:70]}')
