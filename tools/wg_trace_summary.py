"""Summarise tools/wg_trace.py output: workgroups per (xcc, se, cu), K-loop spans, start skew.   usage: wg_trace_summary.py <trace.txt>"""
import collections
import re
import sys

rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"WGTRACE blk (\d+) gemm (\d+) split (\d+) tile (\d+),(\d+) xcc (\d+) se (\d+) cu (\d+) start (\d+) end (\d+) nk (\d+)", ln)
    if m:
        rows.append([int(v) for v in m.groups()])
if not rows:
    sys.exit("no WGTRACE lines")
t0 = min(r[8] for r in rows)
per_cu = collections.Counter((r[5], r[6], r[7]) for r in rows)
print(f"{len(rows)} workgroups on {len(per_cu)} distinct (xcc,se,cu); max per cu {max(per_cu.values())}; histogram {collections.Counter(per_cu.values())}")
spans = sorted(r[9] - r[8] for r in rows)
print(f"K-loop span cycles: min {spans[0]} median {spans[len(spans)//2]} max {spans[-1]}")
starts = sorted(r[8] - t0 for r in rows)
ends = sorted(r[9] - t0 for r in rows)
print(f"start offsets: median {starts[len(starts)//2]} max {starts[-1]};  end offsets: median {ends[len(ends)//2]} max {ends[-1]}")
by_xcc = collections.Counter(r[5] for r in rows)
print("workgroups per xcc:", dict(sorted(by_xcc.items())))
mism = sum(1 for r in rows if r[5] != r[0] % 8)
print(f"blocks whose xcc != blockIdx % 8: {mism}")
late = [r for r in rows if r[8] - t0 > 20000]
print(f"workgroups starting > 20k cycles after the first: {len(late)}")
for r in sorted(rows, key=lambda r: r[8])[-8:]:
    print("  late:", dict(blk=r[0], gemm=r[1], split=r[2], xcc=r[5], se=r[6], cu=r[7], start=r[8] - t0, span=r[9] - r[8], nk=r[10]))
