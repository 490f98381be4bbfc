"""Print the interesting fields of a bench.py JSON line.   usage: bench_brief.py <bench.json>"""
import json
import sys

try:
    d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
    print("ms/step %.4f  value %.0f  %s" % (d["ms_per_step"], d["value"], d["config"]["launch"]))
    for k, v in d.get("roofline_families", {}).items():
        print("  %-10s %6.1f us/launch  frac %.3f (executed %.3f)  mfma_util %s" % (
            k, v["avg_launch_us"], v["frac"], v["frac_executed"], v.get("mfma_util")))
    for o in d.get("other_configs", []):
        print("  ", o["config"][:30], "%.3f ms" % o["ms_per_step"] if "ms_per_step" in o else o.get("error"))
    for v in d.get("dtype_variants", []):
        print("  ", v["dtype"], "%.4f ms" % v["ms_per_step"] if "ms_per_step" in v else v.get("error"))
    c = d.get("cpu_baseline")
    if c:
        print("  cpu %s: %.1f ms/step on %d threads" % (c["kind"], c["ms_per_step"], c["cores"]))
except Exception as e:  # noqa: BLE001
    print("bench.json unreadable:", e)
