"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, separate runs) of bench.py into
per-kernel HBM traffic per launch: markdown table on stdout + profiles/r02_pmc_traffic.json (read by bench.py).

usage: python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json> [code version]
Units / corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> fetch x 2; WRITE_SIZE as reported."""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    n, f = fetch[k]
    w = write.get(k, (0, 0.0))[1]
    rows.append((k, n, f, 2 * f * 1024 / 1e6, w, (2 * f + w) * 1024 / 1e6))
rows.sort(key=lambda r: -r[5] * r[1])
print("| kernel | launches | FETCH_SIZE KB (raw) | fetch MB (x2) | WRITE_SIZE KB | traffic MB/launch |")
print("|---|---|---|---|---|---|")
for k, n, f, fmb, w, t in rows[:28]:
    print(f"| `{k[:110]}` | {n} | {f:.1f} | {fmb:.2f} | {w:.1f} | {t:.2f} |")


def fam(pred):
    sel = [r for r in rows if pred(r[0])]
    tot = sum(r[1] for r in sel)
    return sum(r[5] * r[1] for r in sel) / tot * 1e6 if tot else None


out = {"source": "profiles/r02_pmc_fetch_write.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, fetch x 2 gfx950 correction)"
                 + (", code " + sys.argv[4] if len(sys.argv) > 4 else ""),
       "unit": "bytes per launch (average over the launches of the family)",
       "kernels": {
           "glu_fwd": fam(lambda k: "sg_gemm2<GluFwdEpi" in k),
           "glu_dgrad": fam(lambda k: "sg_gemm2<GluDpreEpi" in k or "GluDgrad0Op" in k),
           "glu_wgrad": fam(lambda k: "sg_gemm2<G2SlabEpi, false, false, true, 128" in k or "sg_gemm2<G2SlabEpi, false, false, true, 64" in k),
       }}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("\n" + json.dumps(out["kernels"]))
