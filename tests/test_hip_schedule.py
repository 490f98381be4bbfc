"""The data-parallel train step's SCHEDULES, proven on a 1-GPU box with a collective that changes data.

One-rank RCCL all-reduces are the identity, so they cannot see a range reduced before its gradients are final or a missing
side -> main stream edge.  `TrainStep(collective_fn=...)` puts an in-stream stand-in wherever the step would call
torch.distributed: here `view *= 2` -- what a second rank holding identical gradients contributes to a SUM -- with world = 2,
so the optimizer's 1 / world makes a CORRECT schedule reproduce the plain single-process step bit for bit, while a
range that is reduced early (the product is overwritten by the late gradient), twice, or not at all changes the update by a
factor of 2.  Checked at the headline shape with the side-stream overlap on and real dropout:
  * one-graph two-range (the default at N > 1)  ==  hipGraph(fwd+bwd) + flat all-reduce + hipGraph(optimizer)  ==  plain step
  * exact mode's two [N,N] collectives captured in the one graph  ==  the same step run eagerly
  * the start-up verification of the one-graph step catches a tail range that is not reduced and falls back
  * the schedule self-check publishes the branch overlap it measured
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPE = dict(N=228, W=12, H=3, multi=5, B=32, T=3000)


def _double(view):
    view.mul_(2.0)


_double.world = 2


def _quadruple(view):          # three more ranks holding identical gradients
    view.mul_(4.0)


_quadruple.world = 4


def _train(steps, collective_fn=None, one_graph=None, exact=False, graph=True, tamper=None, schedule_check=False, shape=SHAPE,
           trace=None):
    """tamper(step): runs right after the TrainStep is built (e.g. `_serial`: the one-stream schedule from the first step on)"""
    from stemgnn_amd import Model, ops
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop
    c = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Model(c["N"], 2, c["W"], c["multi"], horizon=c["H"]).to(dev).train()        # dropout 0.5
    model.set_dropout_seed(99)
    opt = FusedRMSprop(model.parameters(), lr=1e-4, eps=1e-8)
    g = torch.Generator().manual_seed(7)
    series = torch.randn(c["T"], c["N"], generator=g).to(dev)
    total = steps + 1
    hi = (torch.randint(0, c["T"] - c["W"] - c["H"], (total * c["B"],), generator=g) + c["W"]).to(dev)
    world = int(getattr(collective_fn, "world", 2)) if collective_fn is not None else 1
    step = TrainStep(model, opt, c["B"], c["W"], c["H"], c["N"], series=series, world=world, graph=graph, exact=exact,
                     collective_fn=collective_fn, one_graph=one_graph, order_capacity=total * c["B"],
                     schedule_check=schedule_check)
    if tamper is not None:
        tamper(step)
    step.load_order(hi)
    for _ in range(total):
        step.run_next()
        if trace is not None:            # per-step loss and iterator position (diagnostics of a failing comparison)
            torch.cuda.synchronize()
            trace.append((float(step.loss), step.queue.tolist() if step.queue is not None else None))
    torch.cuda.synchronize()
    ops.check_gru_status(dev)
    ops.check_gather_status(dev)
    return opt.flat_p.clone(), step


def test_one_graph_two_range_equals_two_graph_equals_plain_with_a_data_changing_collective():
    steps = 50
    p_plain, s_plain = _train(steps)
    p_two, s_two = _train(steps, _double, one_graph=False)
    p_one, s_one = _train(steps, _double, one_graph=True, schedule_check=True)
    assert s_plain.mode == "hipgraph(whole step)"
    assert s_two.mode.startswith("hipgraph(fwd+bwd)"), s_two.mode
    assert s_one.mode == "hipgraph(whole step incl. rccl all-reduce)", (s_one.mode, s_one.schedule)
    assert s_one._split is not None and s_one.schedule["one_graph_verified"]["ok"], s_one.schedule
    assert "tail_hook_missed" not in s_one.schedule, s_one.schedule
    assert torch.isfinite(p_one).all()
    assert torch.equal(p_one, p_two), float((p_one - p_two).abs().max())
    assert torch.equal(p_one, p_plain), float((p_one - p_plain).abs().max())
    sch = s_one.schedule
    print("schedule:", sch)
    assert sch["checked"], sch
    assert sch["t_serial_ms"] > 0 and sch["side_sum_ms"] > 0 and sch["t_overlap_ms"] > 0


def test_start_up_verification_at_world_4_compares_second_moments():
    """world > 2: the ring's summation order is not the flat form's, so the start-up verification compares the optimizer's
    second moments to 1e-4 of their largest change instead of the parameters bit for bit (RMSprop turns the rounding noise of
    the ~4 000 zero-gradient weights into whole steps).  With the x 4 stand-in the sums are exact: the check must pass, the
    one-graph form must be adopted and reproduce the plain step."""
    p_plain, _ = _train(6, shape=dict(SHAPE, T=800))
    p_one, s_one = _train(6, _quadruple, one_graph=True, shape=dict(SHAPE, T=800))
    v = s_one.schedule["one_graph_verified"]
    assert v["ok"] and v["world"] == 4 and v["compared"].startswith("second moments"), v
    assert v["max_abs_change"] > 0 and v["max_abs_diff"] <= 1e-4 * v["max_abs_change"], v
    assert s_one.mode == "hipgraph(whole step incl. rccl all-reduce)", s_one.mode
    assert torch.equal(p_one, p_plain)


def test_exact_mode_collectives_inside_the_graph_equal_the_eager_step():
    steps = 12
    shape = dict(SHAPE, T=1200)
    p_eager, s_eager = _train(steps, _double, exact=True, graph=False, shape=shape)
    p_graph, s_graph = _train(steps, _double, exact=True, one_graph=True, shape=shape)
    assert s_eager.mode == "eager"
    assert s_graph.mode.startswith("hipgraph(whole step incl. the exact-mode"), (s_graph.mode, s_graph.schedule)
    assert s_graph.schedule["one_graph_verified"]["ok"], s_graph.schedule
    assert torch.equal(p_graph, p_eager), float((p_graph - p_eager).abs().max())


def test_start_up_verification_rejects_a_tail_range_that_is_not_reduced():
    """Negative control of TrainStep._verify_one_graph: the side-branch hook claims the block / fc range but reduces
    nothing -> the captured step applies un-reduced gradients there; the verification must see it and every later step must
    run in the two-graph form, with the parameters of the correct schedule."""
    def tamper(step):
        def fake_tail():
            step._tail_reduced = True
        step.state.block_grads_hook = fake_tail
    steps = 6
    shape = dict(SHAPE, T=800)
    p_ref, _ = _train(steps, _double, one_graph=False, shape=shape)
    p_bad, s_bad = _train(steps, _double, one_graph=True, tamper=tamper, shape=shape)
    v = s_bad.schedule["one_graph_verified"]
    assert not v["ok"] and v["max_abs_diff"] > 0, v
    assert s_bad.mode.startswith("hipgraph(fwd+bwd)"), s_bad.mode
    # step 0 ran eagerly with the tampered hook (nothing can check an eager first step against itself); from the capture on
    # the fallback applies correct gradients -- so the run stays finite and close to the reference run, not equal to it
    assert torch.isfinite(p_bad).all()
    assert float((p_bad - p_ref).abs().max()) < 1e-2


def test_schedule_self_check_reports_branch_overlap_on_the_plain_step():
    p, s = _train(8, schedule_check=True, shape=dict(SHAPE, T=800))
    sch = s.schedule
    print("schedule:", sch)
    assert sch["checked"], sch
    assert sch["recaptures"] <= 3
    # the side branch is ~0.3 ms of kernels; serialising it must cost time, overlapping it must win most of it back
    assert sch["t_serial_ms"] > sch["t_overlap_ms"], sch
    assert sch["branch_overlap"] > 0.3, sch           # a healthy capture measures ~0.56 (engine.LOST_OVERLAP = 0.3)
    p2, _ = _train(8, schedule_check=False, shape=dict(SHAPE, T=800))
    assert torch.equal(p, p2)          # the check runs under snapshot / restore: training is unchanged by it


def _serial(step):
    step.state.overlap = False


def test_serial_graph_equals_the_eager_one_stream_step():
    """The serialised schedule (everything on one stream: what the self-check adopts on a device that gives no branch overlap)
    captured into a hipGraph runs the SAME kernels with the same split counts as the eager one-stream step, so the two must
    agree BIT FOR BIT over several optimizer steps -- and either must agree with the overlapped step to fp32 re-association
    (block 1's weight-gradient launch is sized for the whole chip instead of the CUs the GRU leaves free: another split
    count).  Round 5 saw them 1.3 % apart and called it RMSprop noise; round 6 found the cause: the HIP
    runtime replays a MEMSET NODE of a single-stream captured graph wrongly from the second replay on (pure-torch
    reproducer: tools/diag/graph_memset_probe.py), the arrival counters of block 1's weight-gradient launch were not zero,
    no workgroup saw itself as the last arriver and block 1's gradients were whatever its buffer held.  Every zero fill of the
    step path is a kernel node now (csrc/devattr.h sg_zero_async)."""
    steps = 10
    shape = dict(SHAPE, T=800)
    p_eager, s_eager = _train(steps, graph=False, tamper=_serial, shape=shape)
    p_graph, s_graph = _train(steps, tamper=_serial, shape=shape)
    p_over, s_over = _train(steps, shape=shape)
    assert s_eager.mode == "eager" and s_graph.mode == "hipgraph(whole step)" and s_over.mode == "hipgraph(whole step)"
    assert s_graph.state.overlap is False and s_over.state.overlap is True
    assert torch.isfinite(p_graph).all()
    assert torch.equal(p_graph, p_eager), float((p_graph - p_eager).abs().max())
    assert float((p_graph - p_over).norm() / p_over.norm()) < 1e-6
    assert abs(float(s_graph.loss) - float(s_over.loss)) < 1e-5 * abs(float(s_over.loss))


def test_serial_and_overlapped_graphs_follow_the_oracle_trainer_over_ten_steps():
    """Both captured schedules against the CPU oracle's train loop (oracle.OracleTrainer: zero_grad -> forward -> MSE ->
    backward -> torch RMSprop, models/handler.py:157-166) on the same deterministic weights and batches, dropout 0: the
    per-step losses of ten optimizer steps agree to 2e-5 relative (the single-step parity is ~1e-6; what accumulates is the
    early RMSprop steps' sensitivity to the last bits of small gradients, the same for both schedules)."""
    from oracle import stemgnn_oracle as O
    from stemgnn_amd import Model
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop
    c = dict(SHAPE, T=600)
    steps = 10
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    series = torch.randn(c["T"], c["N"], generator=g)
    hi = torch.randint(0, c["T"] - c["W"] - c["H"], (steps * c["B"],), generator=g) + c["W"]
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    tr = O.OracleTrainer(c["N"], c["W"], c["multi"], c["H"], lr=1e-4, seed=3, dropout_rate=0.0)
    ref = []
    for k in range(steps):
        idx = hi[k * c["B"]:(k + 1) * c["B"]]
        x = torch.stack([series[i - c["W"]:i] for i in idx.tolist()])
        y = torch.stack([series[i:i + c["H"]] for i in idx.tolist()])
        ref.append(tr.step(x, y))
    for serial in (True, False):
        model = Model(c["N"], 2, c["W"], c["multi"], horizon=c["H"], dropout_rate=0.0)
        model.load_state_dict(O.det_state_dict(c["N"], c["W"], c["multi"], c["H"], seed=3))
        model = model.to(dev).train()
        opt = FusedRMSprop(model.parameters(), lr=1e-4, eps=1e-8)
        step = TrainStep(model, opt, c["B"], c["W"], c["H"], c["N"], series=series.to(dev), world=1, graph=True,
                         order_capacity=steps * c["B"], schedule_check=False)
        if serial:
            step.state.overlap = False
        step.load_order(hi.to(dev))
        got = []
        for _ in range(steps):
            step.run_next()
            torch.cuda.synchronize()
            got.append(float(step.loss))
        assert step.mode == "hipgraph(whole step)" and step.state.overlap is (not serial)
        worst = max(abs(a - b) / abs(b) for a, b in zip(got, ref))
        print("serial" if serial else "overlapped", "worst relative loss difference to the oracle over", steps, "steps:", worst)
        assert worst < 2e-5, (serial, got, ref)


def test_a_capture_without_branch_overlap_is_re_captured_and_then_serialised(monkeypatch):
    """The 1-in-6 box of round 4 (both branches on one hardware queue: 1.60 ms instead of 1.23, slower than the serialised
    step's 1.47) cannot be provoked at will, so its TIMINGS are: every timing of an overlapped capture is reported 1.5 x
    the serialised step's.  The self-check must re-capture three times, then adopt the serialised graph -- after CHECKING
    one replay of it bit for bit against the eager one-stream step (`schedule["adopted_verified"]`) --, say so in `schedule`
    / `mode` -- and training must go on: same kernels on the same data, but block 1's weight-gradient launch is sized for
    the whole chip instead of the CUs the GRU leaves free (another split count = another fp32 summation order), so the run
    agrees with the overlapped one to fp32 re-association: measured 3e-8 of the parameter norm after 9 steps, losses to
    1e-7 (round 5 accepted 3 % here; the cause was a mis-replayed memset node, see the test above)."""
    from stemgnn_amd import engine
    real = engine._time_replays
    calls = {"n": 0, "serial": None}

    def fake(replay, n=10):
        calls["n"] += 1
        ms = real(replay, n)
        if calls["n"] == 2:
            calls["serial"] = ms
        return ms if calls["n"] <= 2 else 1.5 * calls["serial"]       # 1: side branch alone, 2: serialised step, 3+: overlapped
    monkeypatch.setattr(engine, "_time_replays", fake)
    shape = dict(SHAPE, T=800)
    tr, tr2 = [], []
    p, s = _train(8, schedule_check=True, shape=shape, trace=tr)
    sch = s.schedule
    print("schedule:", sch)
    assert sch["checked"] and sch["recaptures"] == 3 and sch["side_branch_serialised"], sch
    assert len(sch["t_overlap_ms_per_capture"]) == 4
    assert "side branch serialised" in s.mode and s.state.overlap is False
    assert sch["adopted_verified"]["ok"] and sch["adopted_verified"]["against"].startswith("eager one-stream"), sch
    monkeypatch.setattr(engine, "_time_replays", real)
    p2, s2 = _train(8, schedule_check=False, shape=shape, trace=tr2)
    assert s2.mode == "hipgraph(whole step)"
    assert torch.isfinite(p).all()
    import json
    import os
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "schedule_adopted_trace.json"), "w") as f:
            json.dump({"adopted": tr, "overlapped": tr2, "schedule": {k: v for k, v in sch.items() if k != "error"},
                       "max_abs": float((p - p2).abs().max()), "rel_norm": float((p - p2).norm() / p2.norm())}, f)
    assert float((p - p2).norm() / p2.norm()) < 1e-6, (tr, tr2)
    for (la, _), (lb, _) in zip(tr, tr2):
        assert abs(la - lb) < 1e-5 * abs(lb), (tr, tr2)
