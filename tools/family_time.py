"""The three GLU GEMM families timed in isolation (what bench.py's `roofline_families` reports), as a stand-alone command
so that a rocprofv3 kernel trace of it holds ONLY isolated launches (the in-step launches are profiled separately with
`bench.py --no-roofline`).   usage: python tools/family_time.py [N,W,H,multi,B]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = dict(bench.WORKLOAD)
if len(sys.argv) > 1:
    cfg = dict(zip(("N", "W", "H", "multi", "B"), (int(v) for v in sys.argv[1].split(","))))
main, rows = bench.roofline_objects(cfg)
print(json.dumps({k: {kk: v[kk] for kk in ("avg_launch_us", "sum_us_per_step", "frac", "frac_executed")} for k, v in rows.items()}))
