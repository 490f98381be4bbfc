"""Summarise the rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + SQ_WAVE_CYCLES +
GRBM_GUI_ACTIVE -- kernel-trace only, separate runs) of bench.py into per-kernel HBM traffic per launch and MFMA
utilisation: markdown table on stdout + a JSON (profiles/rNN_pmc_traffic.json, read by bench.py).

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 256 CUs x 4 SIMDs): the share of the chip's
matrix-pipe cycles (per SIMD) that were busy while the kernel ran.  SQ_VALU_MFMA_BUSY_CYCLES is an exact instruction census
(64 cycles per v_mfma_f32_32x32x2_f32, summed over every SIMD: GLU forward = 496 128 MFMAs per launch x 64 = the counter
to the digit).  The denominator uses the dispatch's own Start/End timestamps of the counter pass: GRBM_GUI_ACTIVE on this
stack is the SUM over the 8 XCDs and carries ~8 us of per-dispatch set-up (GUI / 8 / duration -> 2.5 cycles/ns for a 100 us
kernel, 4.9 for a 7 us one), so the gfx94x MfmaUtil formula (busy / (GUI x CUs x 4)) under-reads by 8x and more; the
GUI-based figure (with the / 8) is printed beside it for comparison.

usage: python tools/pmc_summary.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> <dir MFMA pass | -> <out.json> [code version]
Units / corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KB;
on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> fetch x 2; WRITE_SIZE as reported."""
import csv
import glob
import json
import sys
from collections import defaultdict


CLOCK_GHZ = 2.4


def collect(d, counter, duration=False):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            # round 6: the fused forward's warm-up launch (eight workgroups on a dummy buffer, stemgnn_spectral_glu_fwd_warm) is
            # not a launch of the family: it would pull the per-launch averages down
            if "sg_glu_fused_fwd" in r["Kernel_Name"] and int(r["Grid_Size"]) <= 8 * 256:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) if duration else float(r["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
mdir = sys.argv[3]
OUT_JSON, VERSION = sys.argv[4], (sys.argv[5] if len(sys.argv) > 5 else None)
mfma = collect(mdir, "SQ_VALU_MFMA_BUSY_CYCLES") if mdir != "-" else {}
gui = collect(mdir, "GRBM_GUI_ACTIVE") if mdir != "-" else {}
sqbusy = collect(mdir, "SQ_BUSY_CYCLES") if mdir != "-" else {}
dur = collect(mdir, "SQ_VALU_MFMA_BUSY_CYCLES", duration=True) if mdir != "-" else {}      # ns per dispatch (counter pass)
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
    sel = [k for k in mfma if pred(k) and k in dur and dur[k][1] > 0]
    if not sel:
        return None
    busy = sum(mfma[k][0] * mfma[k][1] for k in sel)
    cyc = sum(dur[k][0] * dur[k][1] * CLOCK_GHZ for k in sel)
    return busy / (cyc * 256 * 4)


if mfma:
    print("\n| kernel | launches | MFMA_BUSY cycles / launch | duration us (counter pass) | GRBM_GUI_ACTIVE / launch | "
          "MFMA util (duration x 2.4 GHz x 1024 SIMDs) | via GUI / 8 |")
    print("|---|---|---|---|---|---|---|")
    for k in sorted(mfma, key=lambda k: -mfma[k][0] * mfma[k][1])[:20]:
        g = gui.get(k, (0, 0.0))[1]
        dn = dur.get(k, (0, 0.0))[1]
        print(f"| `{k[:110]}` | {mfma[k][0]} | {mfma[k][1]:.0f} | {dn / 1e3:.1f} | {g:.0f} | "
              f"{(mfma[k][1] / (dn * CLOCK_GHZ * 1024) if dn else 0):.3f} | {(mfma[k][1] / (g / 8 * 1024) if g else 0):.3f} |")

FAMS = {
    "glu_fwd": lambda k: "GluFwdEpi" in k or "sg_glu_fused_fwd" in k,            # round 4: one fused launch per block
    "glu_dgrad": lambda k: "GluDpreEpi" in k or "GluDgrad0Op" in k or "sg_glu_fused_dgrad" in k,
    "glu_wgrad": lambda k: "G2SlabEpi, false, false, true, 128" in k or "G2SlabEpi, false, false, true, 64" in k or "sg_wgrad" in k,
    # round 6: the recurrences (bench.py's `roofline` names the GRU backward: the largest critical-path family)
    "gru_fwd": lambda k: "gru_fwd_cluster" in k or "gru_fwd_wide" in k,
    "gru_bwd": lambda k: "gru_bwd_cluster" in k or "gru_bwd_wide" in k,
}
out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES / (dispatch duration x 2.4 GHz x 1024 SIMDs))"
                 + (", code " + VERSION if VERSION else ""),
       "mfma_util": {f: util(p) for f, p in FAMS.items()},
       "unit": "bytes per launch (average over the launches of the family)",
       "kernels": {f: fam(p) for f, p in FAMS.items()}}
# "kernels" = bytes per LAUNCH averaged over the family's launches as profiled; bench.py rescales to its own launch count
json.dump(out, open(OUT_JSON, "w"), indent=1)
print("\n" + json.dumps({"traffic": out["kernels"], "mfma_util": out["mfma_util"]}))
