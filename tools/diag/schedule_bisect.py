"""Bisect of the serialised-schedule divergence: the same 9 optimizer steps through every schedule of the
train step, with the flat parameters and RMSprop second moments kept per step, compared pairwise and BY PARAMETER RANGE at
the first step where two runs part.

    python tools/diag/schedule_bisect.py [out.json] [dropout]

kinds:  overlap        hipGraph, side branch beside the chain (the default step)
        overlap_eager  the same step, no graph
        serial_eager   no graph, everything on one stream (state.overlap = False from the start)
        serial_graph   hipGraph of the one-stream step (state.overlap = False from the start)
        adopted        the schedule self-check with injected timings: 3 re-captures, then the serialised graph is adopted
`diagnose()` is importable: tests/test_hip_schedule.py calls it when its two runs part by more than the bound, so the same
table exists for the in-tier context (end of the whole -m gpu process), where round 5 saw 1.3 % against 2e-4 in a fresh process.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

SHAPE = dict(N=228, W=12, H=3, multi=5, B=32, T=800)


def run(kind, steps=8, dropout=0.5, shape=SHAPE, fuse_zero=True):
    from stemgnn_amd import Model, engine, ops
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop
    c = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Model(c["N"], 2, c["W"], c["multi"], horizon=c["H"], dropout_rate=dropout).to(dev).train()
    model.set_dropout_seed(99)
    opt = FusedRMSprop(model.parameters(), lr=1e-4, eps=1e-8, fuse_zero_grad=fuse_zero)
    g = torch.Generator().manual_seed(7)
    series = torch.randn(c["T"], c["N"], generator=g).to(dev)
    total = steps + 1
    hi = (torch.randint(0, c["T"] - c["W"] - c["H"], (total * c["B"],), generator=g) + c["W"]).to(dev)
    graph = kind in ("overlap", "serial_graph", "adopted")
    step = TrainStep(model, opt, c["B"], c["W"], c["H"], c["N"], series=series, world=1, graph=graph,
                     order_capacity=total * c["B"], schedule_check=(kind == "adopted"))
    if kind in ("serial_eager", "serial_graph"):
        step.state.overlap = False
    real = engine._time_replays
    calls = {"n": 0, "serial": None}

    def fake(replay, n=10):
        calls["n"] += 1
        ms = real(replay, n)
        if calls["n"] == 2:
            calls["serial"] = ms
        return ms if calls["n"] <= 2 else 1.5 * calls["serial"]
    if kind == "adopted":
        engine._time_replays = fake
    names = [(n, opt.bucket.offset_of(p), p.numel()) for n, p in model.named_parameters() if p.requires_grad]
    out = dict(kind=kind, loss=[], seed=[], p=[], v=[], g=[], names=names)
    try:
        step.load_order(hi)
        for _ in range(total):
            step.run_next()
            torch.cuda.synchronize()
            out["loss"].append(float(step.loss))
            out["seed"].append(model._seed.tolist() if model._seed is not None else None)
            out["p"].append(opt.flat_p.detach().cpu().clone())
            out["v"].append(opt.square_avg.detach().cpu().clone())
            if not fuse_zero:       # the step's gradients are still in the flat bucket (zeroed at the START of the next step)
                out["g"].append(opt.bucket.flat.detach().cpu().clone())
    finally:
        engine._time_replays = real
    ops.check_gru_status(dev)
    ops.check_gather_status(dev)
    out["mode"] = step.mode
    out["schedule"] = {k: v for k, v in step.schedule.items() if k != "error"}
    return out


def compare(a, b, alpha=0.99):
    """First step at which the two runs' parameters differ, and there: per parameter range the count of differing elements,
    max |dp|, and the relative difference of the gradient magnitude recovered from the second moments
    (g^2 = (v_k - alpha v_{k-1}) / (1 - alpha))."""
    res = dict(a=a["kind"], b=b["kind"], losses_a=a["loss"], losses_b=b["loss"], seeds_equal=a["seed"] == b["seed"])
    first = None
    for k in range(len(a["p"])):
        if not (torch.equal(a["p"][k], b["p"][k]) and torch.equal(a["v"][k], b["v"][k])):
            first = k
            break
    res["first_differing_step"] = first
    res["rel_norm_last"] = float((a["p"][-1] - b["p"][-1]).norm() / b["p"][-1].norm())
    if first is None:
        return res
    k = first
    va0 = a["v"][k - 1] if k > 0 else torch.zeros_like(a["v"][0])
    vb0 = b["v"][k - 1] if k > 0 else torch.zeros_like(b["v"][0])
    ga = ((a["v"][k].double() - alpha * va0.double()) / (1 - alpha)).clamp_min(0).sqrt()
    gb = ((b["v"][k].double() - alpha * vb0.double()) / (1 - alpha)).clamp_min(0).sqrt()
    rows = []
    for name, off, n in a["names"]:
        sl = slice(off, off + n)
        dp = (a["p"][k][sl] - b["p"][k][sl]).abs()
        ndiff = int((dp > 0).sum())
        dv = int((a["v"][k][sl] != b["v"][k][sl]).sum())
        if ndiff == 0 and dv == 0:
            continue
        gmax = float(gb[sl].max())
        row = dict(name=name, numel=n, p_differ=ndiff, v_differ=dv, max_dp=float(dp.max()),
                   g_max=gmax, max_dg_over_gmax=float((ga[sl] - gb[sl]).abs().max() / max(gmax, 1e-300)),
                   g_rel_l2=float((ga[sl] - gb[sl]).norm() / max(float(gb[sl].norm()), 1e-300)))
        if a["g"] and b["g"]:      # the gradients themselves: zero? the previous step's (stale)? something else?
            xa, xb = a["g"][k][sl].double(), b["g"][k][sl].double()
            nb = max(float(xb.norm()), 1e-300)
            row.update(grad_norm_a=float(xa.norm()), grad_norm_b=float(xb.norm()), grad_diff_rel=float((xa - xb).norm()) / nb,
                       nonfinite_a=int((~torch.isfinite(xa)).sum()))
            if k > 0:
                row["a_vs_previous_step_b_rel"] = float((xa - b["g"][k - 1][sl].double()).norm()) / nb
                row["a_vs_previous_step_a_rel"] = float((xa - a["g"][k - 1][sl].double()).norm()) / nb
        rows.append(row)
    res["ranges_at_first_difference"] = rows
    return res


def diagnose(out_path=None, dropout=0.5, kinds=("overlap", "overlap_eager", "serial_eager", "serial_graph", "adopted"),
             pairs=(("overlap", "overlap_eager"), ("serial_eager", "serial_graph"), ("serial_graph", "adopted"),
                    ("overlap", "serial_eager"), ("overlap", "adopted")), fuse_zero=True):
    runs = {k: run(k, dropout=dropout, fuse_zero=fuse_zero) for k in kinds}
    report = dict(dropout=dropout, modes={k: r["mode"] for k, r in runs.items()},
                  losses={k: r["loss"] for k, r in runs.items()},
                  schedule={k: r["schedule"] for k, r in runs.items() if r["schedule"].get("checked")},
                  pairs=[compare(runs[x], runs[y]) for x, y in pairs if x in runs and y in runs])
    if out_path:
        with open(out_path, "w") as f:
            json.dump(report, f, indent=1)
    return report


def brief(report):
    for k, l in report["losses"].items():
        print(f"{k:14s} {report['modes'][k][:44]:44s} " + " ".join(f"{v:.8f}" for v in l))
    for c in report["pairs"]:
        print(f"-- {c['a']} vs {c['b']}: first differing step {c['first_differing_step']}, seeds equal {c['seeds_equal']}, "
              f"rel norm after the last step {c['rel_norm_last']:.3e}")
        for r in c.get("ranges_at_first_difference", [])[:80]:
            extra = ""
            if "grad_norm_a" in r:
                extra = (f" |ga| {r['grad_norm_a']:.3e} |gb| {r['grad_norm_b']:.3e} |ga-gb|/|gb| {r['grad_diff_rel']:.2e} nonfinite {r['nonfinite_a']}"
                         f" vs prev b {r.get('a_vs_previous_step_b_rel', -1):.2e} vs prev a {r.get('a_vs_previous_step_a_rel', -1):.2e}")
            print(f"     {r['name']:52s} n={r['numel']:7d} p!= {r['p_differ']:7d} v!= {r['v_differ']:7d} max dp {r['max_dp']:.2e} "
                  f"max dg/gmax {r['max_dg_over_gmax']:.2e} g rel l2 {r['g_rel_l2']:.2e}" + extra)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else None
    drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    if len(sys.argv) > 3 and sys.argv[3] == "grads":     # gradients kept per step (zeroing NOT fused into the optimizer kernel)
        brief(diagnose(path, dropout=drop, kinds=("serial_eager", "serial_graph"), pairs=(("serial_graph", "serial_eager"),),
                       fuse_zero=False))
    else:
        brief(diagnose(path, dropout=drop))
