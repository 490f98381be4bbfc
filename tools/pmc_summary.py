"""Summarise the rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + SQ_WAVE_CYCLES +
GRBM_GUI_ACTIVE -- kernel-trace only, separate runs) of bench.py into per-kernel HBM traffic per launch and MFMA
utilisation: markdown table on stdout + a JSON (profiles/rNN_pmc_traffic.json, read by bench.py).

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs): the share of the chip's matrix-pipe
cycles (per SIMD) that were busy while the kernel ran (the gfx94x MfmaUtil formula; ROCm 7.2 has no gfx950 section).

usage: python tools/pmc_summary.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> <dir MFMA pass | -> <out.json> [code version]
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
mdir = sys.argv[3]
OUT_JSON, VERSION = sys.argv[4], (sys.argv[5] if len(sys.argv) > 5 else None)
mfma = collect(mdir, "SQ_VALU_MFMA_BUSY_CYCLES") if mdir != "-" else {}
gui = collect(mdir, "GRBM_GUI_ACTIVE") if mdir != "-" else {}
sqbusy = collect(mdir, "SQ_BUSY_CYCLES") if mdir != "-" else {}
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


def util(pred):
    sel = [k for k in mfma if pred(k) and k in gui and gui[k][1] > 0]
    if not sel:
        return None
    busy = sum(mfma[k][0] * mfma[k][1] for k in sel)
    act = sum(gui[k][0] * gui[k][1] for k in sel)
    return busy / (act * 256 * 4)


if mfma:
    print("\n| kernel | launches | MFMA_BUSY cycles / launch | GRBM_GUI_ACTIVE / launch | SQ_BUSY_CYCLES / launch | MFMA util |")
    print("|---|---|---|---|---|---|")
    for k in sorted(mfma, key=lambda k: -mfma[k][0] * mfma[k][1])[:20]:
        g = gui.get(k, (0, 0.0))[1]
        print(f"| `{k[:110]}` | {mfma[k][0]} | {mfma[k][1]:.0f} | {g:.0f} | {sqbusy.get(k, (0, 0.0))[1]:.0f} | "
              f"{(mfma[k][1] / (g * 1024) if g else 0):.3f} |")

FAMS = {
    "glu_fwd": lambda k: "GluFwdEpi" in k,
    "glu_dgrad": lambda k: "GluDpreEpi" in k or "GluDgrad0Op" in k,
    "glu_wgrad": lambda k: "G2SlabEpi, false, false, true, 128" in k or "G2SlabEpi, false, false, true, 64" in k or "sg_wgrad" in k,
}
out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024))"
                 + (", code " + VERSION if VERSION else ""),
       "mfma_util": {f: util(p) for f, p in FAMS.items()},
       "unit": "bytes per launch (average over the launches of the family)",
       "kernels": {f: fam(p) for f, p in FAMS.items()}}
# "kernels" = bytes per LAUNCH averaged over the family's launches as profiled; bench.py rescales to its own launch count
json.dump(out, open(OUT_JSON, "w"), indent=1)
print("\n" + json.dumps({"traffic": out["kernels"], "mfma_util": out["mfma_util"]}))
