"""RCCL readiness on a 1-GPU box (missing 3): the `nccl` (= RCCL) backend is initialised with ONE rank under
the same launcher the driver uses for N > 1, the flat gradient all-reduce really goes through dist.all_reduce, the
collective form of the train step (what N > 1 runs) reproduces the plain step bit for bit, whether a collective can be
captured inside a hipGraph on this stack is PROBED and printed, and bench.py's launcher branch runs end to end."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script_args, timeout=600):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:                                  # keep the whole transcript (the assertion message is truncated)
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, f"rccl_launch_failed_{port}.log"), "w") as f:
                f.write("CMD " + " ".join(cmd) + "\n---- stdout\n" + p.stdout + "\n---- stderr\n" + p.stderr)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    return json.loads(lines[-1])


def test_rccl_single_rank_group_collectives_and_graph_capture_probe():
    res = _launch([os.path.join(ROOT, "tests", "helpers", "rccl_probe.py")])
    print("RCCL probe:", json.dumps(res))
    out_dir = os.path.join(ROOT, "gpurun_out")          # keep the probe's answer (pytest -q swallows stdout)
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "rccl_probe.json"), "w") as f:
            json.dump(res, f, indent=1)
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["allreduce_world"] == 1 and res["allreduce_unchanged"]
    # the wait ahead of a capture ended on an observed state ("drained": the flight recorder lists no unretired collective)
    # or, where the recorder is off, on the fixed wait ("sleep") -- never on its time-out
    assert str(res.get("watchdog_drain")).startswith("drained") or res.get("watchdog_drain") == "sleep", res
    assert str(res.get("watchdog_drain_last_capture")).startswith("drained") or res.get("watchdog_drain_last_capture") == "sleep", res
    assert res["mode_collective"].startswith("hipgraph(fwd+bwd) + rccl all-reduce"), res
    assert res["collective_equals_plain"], res
    if res["graph_capture_allreduce"]:                 # recorded either way; asserted only where the stack supports it
        assert res["mode_one_graph"] == "hipgraph(whole step incl. rccl all-reduce)", res
        assert res["one_graph_equals_plain"], res
        assert res["mode_exact_one_graph"].startswith("hipgraph(whole step incl. the exact-mode"), res
        assert res["exact_one_graph_max_rel"] < 1e-5, res


def test_bench_launcher_branch_with_one_rank():
    """`bench.py --gpus 1` under torch.distributed.run (WORLD_SIZE=1): process group, collective train step, barriers and
    the max-over-ranks timing all execute; the JSON line has the contract's keys."""
    out = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-roofline"])
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 0
    assert "rccl all-reduce" in out["config"]["launch"], out["config"]
    assert out["scaling"] == "weak" and out["unit"] == "forecast-steps/s"
