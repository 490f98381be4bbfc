#!/usr/bin/env python
"""bench.py -- forecast-steps/sec (train) of the StemGNN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one optimizer step of the drop-in model on one synthetic PEMS07-shaped batch
(N=228, W=12, H=3, multi=5, per-GPU batch 32 -- BASELINE.json configs[1]): zero_grad -> forward ->
MSELoss -> backward -> (flat grad all-reduce over RCCL when N>1) -> RMSprop(lr=1e-4, eps=1e-8) step,
exactly the reference's loop body (models/handler.py:157-165); the z-scored series is resident in HBM and the
batch windows are gathered from it by index inside the step (stemgnn_amd.engine.TrainStep, the same object
stemgnn_amd.handler.train drives).
Weak scaling: every rank trains on its own 32-sample batch ("replicas with a local graph").

Rank 0 prints ONE JSON line with the throughput, a `roofline` object for the dominant kernel
(timed live with HIP events on the launch stream) and, at N=1, a `cpu_baseline` object: the CPU oracle
port of the same train step timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOAD = dict(N=228, W=12, H=3, multi=5, B=32)      # PEMS07 shape, BASELINE.json configs[1]
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
ROOFLINE_TRAFFIC_BYTES = 36.3e6                        # profiles/r01_v7_pmc_fetch_write.md (dominant kernel, per launch)


def glu_fwd_flops(B, N, W, multi):
    """Algorithmic FLOPs of one stemgnn_spectral_glu_fwd call (3 GEMM launches, both branches), dense count
    of SURVEY 8d: 2*M*(C0*2C + 2C*C + 2C*C) per branch with C0 = 4W, C = 4*W*multi."""
    M, C0, C = B * N, 4 * W, 4 * W * multi
    return 2 * (2.0 * M * (C0 * 2 * C + C * 2 * C + C * 2 * C))


def time_dominant_kernel(cfg, iters=20):
    """Average launch duration of the dominant roofline kernel (sg_gemm2<GluFwdEpi>: the 3 launches of one
    stemgnn_spectral_glu_fwd call), HIP events on the launch stream."""
    from stemgnn_amd import _lib

    lib = _lib.load()
    B, N, W, multi = cfg["B"], cfg["N"], cfg["W"], cfg["multi"]
    dev = torch.device("cuda")
    packed = torch.randn(lib.stemgnn_packed_floats(W, multi), device=dev) * 0.05
    saved = torch.randn(lib.stemgnn_saved_floats(B, N, W, multi), device=dev)
    st = torch.cuda.current_stream()

    def run():
        _lib.check(lib.stemgnn_spectral_glu_fwd(packed.data_ptr(), saved.data_ptr(), B, N, W, multi, st.cuda_stream),
                   "spectral_glu_fwd")

    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        run()
    e1.record(st)
    e1.synchronize()
    launches = 3 * iters
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches
    flops_per_launch = glu_fwd_flops(B, N, W, multi) / 3.0
    return avg_s, flops_per_launch


def cpu_baseline(cfg, budget_s=20.0):
    """The CPU oracle port of the same train step on this box's host cores (kind='port').

    Thread count: the oracle is many small ops (a 228-step GRU, 228x228 softmax ...); on a many-core host
    using every core is pathologically slow (OpenMP fork/join per tiny op), so a short calibration picks the
    fastest of {8, 16, 32, 64} threads (<= cpu_count) and `cores` reports what was actually used."""
    from oracle.stemgnn_oracle import OracleTrainer

    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(cfg["B"], cfg["W"], cfg["N"], generator=g)
    y = torch.randn(cfg["B"], cfg["H"], cfg["N"], generator=g)
    tr = OracleTrainer(cfg["N"], cfg["W"], cfg["multi"], cfg["H"], lr=1e-4, seed=0, dropout_rate=0.5)
    best_t, best_n = None, None
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        tr.step(x, y)                                   # warm-up at this thread count
        t0 = time.perf_counter()
        tr.step(x, y)
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if dt > 5.0:
            break
    torch.set_num_threads(best_n)
    tr.step(x, y)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(x, y)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 100:
            break
    sps = n / el
    return {"value": cfg["B"] * cfg["H"] * sps, "unit": "forecast-steps/s", "cores": best_n, "kind": "port",
            "ms_per_step": 1e3 / sps, "host_cpus": ncpu,
            "sample": f"{n} train steps of the torch-CPU oracle port (same shape, batch {cfg['B']}, fp32, "
                      f"RMSprop, dropout 0.5), {best_n} threads (fastest of a 8/16/32/64 calibration) on a "
                      f"{ncpu}-cpu host"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from stemgnn_amd import Model
    from stemgnn_amd.distributed import broadcast_parameters
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop

    cfg = dict(WORKLOAD)
    if os.environ.get("STEMGNN_BENCH_WORKLOAD"):     # "N,W,H,multi,B": other BASELINE configs (reported in DESIGN.md only;
        vals = [int(v) for v in os.environ["STEMGNN_BENCH_WORKLOAD"].split(",")]   # the default line stays configs[1])
        cfg = dict(zip(("N", "W", "H", "multi", "B"), vals))
    torch.manual_seed(0)
    model = Model(cfg["N"], 2, cfg["W"], cfg["multi"], horizon=cfg["H"])      # defaults: dropout 0.5, leaky 0.2
    model.to(dev).train()
    broadcast_parameters(model)
    # same arithmetic as the driver's torch.optim.RMSprop (handler.py:127), one fused kernel (+ grad zeroing)
    opt = FusedRMSprop(model.parameters(), lr=1e-4, alpha=0.99, eps=1e-8)
    # synthetic z-scored series resident in HBM (PEMS07 length), windows gathered by index inside the step
    T = 12672
    g = torch.Generator().manual_seed(1234 + rank)
    series = torch.randn(T, cfg["N"], generator=g).to(dev)
    n_windows = T - cfg["W"] - cfg["H"] + 1
    total = args.steps + args.warmup + 1
    epochs = -(-total * cfg["B"] // n_windows)                       # shuffled passes over the windows, back to back
    order = torch.cat([torch.randperm(n_windows, generator=g) for _ in range(epochs)])[: total * cfg["B"]]
    hi_all = (order + cfg["W"]).to(dev).view(total, cfg["B"])        # window-end rows (ForecastDataset.x_end_idx)

    # TrainStep = the driver's loop body (stemgnn_amd/handler.py train): window gather -> zero_grad -> forward ->
    # MSE -> backward (gradients written in place into the flat bucket, block weight-gradient GEMMs overlapping the
    # GRU recurrence on a side stream) -> [RCCL all-reduce] -> RMSprop; captured into hipGraph(s) after the first step
    stepper = TrainStep(model, opt, cfg["B"], cfg["W"], cfg["H"], cfg["N"], series=series, world=world,
                        graph=not args.no_graph)
    stepper.run_indices(hi_all[0])          # eager step (lazy init of tables, seed, state) + graph capture
    torch.cuda.synchronize()
    mode = stepper.mode
    it = iter(range(1, total))

    def step():
        stepper.run_indices(hi_all[next(it)])

    loss_buf = stepper.loss
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss_buf.item())
    if not (final_loss == final_loss) or final_loss > 1e6:
        raise SystemExit(f"training diverged: loss={final_loss}")

    ms_per_step = elapsed / args.steps * 1e3
    value = world * cfg["B"] * cfg["H"] / (elapsed / args.steps)
    out = {
        "metric": "forecast-steps/sec (train)", "value": value, "unit": "forecast-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("PEMS07-shape " if cfg == WORKLOAD else "") +
                               f"N={cfg['N']} W={cfg['W']} H={cfg['H']} multi={cfg['multi']} stack=2, batch {cfg['B']} per GPU, "
                               "train step (window gather+fwd+MSE+bwd+RMSprop), dropout 0.5", "global_batch": world * cfg["B"],
                   "per_gpu_batch": cfg["B"], "parallelism": f"dp{world}", "launch": mode},
        "final_loss": final_loss,
    }
    if rank == 0:
        avg_s, flops = time_dominant_kernel(cfg)
        ach = flops / avg_s / 1e12
        out["roofline"] = {"kernel": "sg_gemm2<GluFwdEpi,true,false,true,64> (spectral GLU forward GEMM, 64x128x16 tiles, "
                                     "v_mfma_f32_32x32x2_f32, exact fp32)",
                           "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / FP32_MFMA_PEAK_TFLOPS, "avg_launch_us": avg_s * 1e6,
                           "flops_per_launch": flops,
                           # HBM bytes per launch from the PMC passes in profiles/r01_v7_pmc_fetch_write.md
                           # (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), not re-measured live
                           "traffic": ROOFLINE_TRAFFIC_BYTES}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
