#!/usr/bin/env python
"""bench.py -- forecast-steps/sec (train) of the StemGNN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment) it is
one rank; started plainly with --gpus N it re-launches itself through torch.distributed.run with N ranks on 127.0.0.1.

A "step" is one optimizer step of the drop-in model on one synthetic PEMS07-shaped batch
(N=228, W=12, H=3, multi=5, per-GPU batch 32 -- BASELINE.json configs[1]): window gather from the HBM-resident series
-> zero_grad -> forward -> MSELoss -> backward -> (flat grad all-reduce over RCCL when N>1) -> RMSprop(lr=1e-4,
eps=1e-8), the reference's loop body (models/handler.py:157-165) as stemgnn_amd.engine.TrainStep runs it (one hipGraph).
Weak scaling: every rank trains on its own 32-sample batch ("replicas with a local graph").
Arithmetic (`--dtype`, `dtype` in the line): default `f32` -- exact fp32, the reference's arithmetic
(data_loader/forecast_dataloader.py:61-62) and the library's default.  BASELINE.json configs[1] also names bf16: `dtype_variants`
carries the same step with the GLU forward and data-gradient products as split-bf16 on the bf16 matrix pipe inside the fused
kernels (`bf16x2`, csrc/glu_fused_bf16.h; <= 3e-5 model-level error against north_star's 1e-4 bar) with its own `roofline`
objects, and `bf16x3`; `--dtype bf16x2` makes that the main line instead.

Rank 0 prints ONE JSON line: the throughput, a `roofline` object for the MFMA GEMM family with the largest summed GPU
time per step (each family timed live with HIP events on the launch stream; `roofline_families` lists all of them),
at N=1 a `cpu_baseline` object (the real reference when /root/reference is mounted, else the oracle port, timed on this
box's host cores) and `other_configs`: short single-GPU runs of the per-GPU shards of BASELINE.json configs[0],[2],[3],[4],
`dtype_variants` (the other arithmetics of the GLU products), `spectral_variants` (the eigensolver route) and `mae_vs_ref` (the
committed reference training run replayed: validation MAE beside the reference's).  The side sections run in CHILD
processes (`--section`), so a fault in one of them costs that section, never the headline; the headline object is also
written to stderr as soon as it exists.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(N=228, W=12, H=3, multi=5, B=32)      # PEMS07 shape, BASELINE.json configs[1]
# Arithmetic of the headline line: exact fp32 -- the reference's arithmetic and the library default (round 5's line was bf16x2;
# VERDICT r5 / ADVICE r5: the headline measures what a user gets by default).  BASELINE.json configs[1] names "bf16/fp32": the
# split-bf16 setting (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation, on v_mfma_f32_32x32x16_bf16 inside the fused
# kernels; model-level error <= 3e-5) is `dtype_variants[0]` of every line, with its own roofline objects.
DEFAULT_DTYPE = "f32"
DTYPE_NOTE = {
    "f32": "exact fp32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): the reference's arithmetic, the library default",
    "bf16x2": "GLU forward, data-gradient AND (round 6) weight-gradient products as split-bf16 (3 bf16 products per fp32 product, "
              "~2^-16 relative) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; GRU, attention, graph products, heads' forward / "
              "data gradients, optimizer exact fp32; model-level error <= 3e-5 norm-relative against the 1e-4 parity bar",
    "bf16x3": "GLU layers 1-2 forward + d(pre-activation) products as 6-term split-bf16 (fp32 class) on the per-layer kernels",
}
# per-GPU shards of the other BASELINE.json configs (global batch / 8 GPUs for configs[3], [4])
OTHER_CONFIGS = [
    ("configs[0] ECG shape", dict(N=140, W=12, H=3, multi=5, B=32)),
    ("configs[2] PEMS03 shape", dict(N=358, W=12, H=3, multi=5, B=32)),
    ("configs[3] N=1024, global batch 64 on 8 GPUs -> per-GPU shard 8", dict(N=1024, W=12, H=3, multi=5, B=8)),
    ("configs[4] N=2048 W=48 H=12, global batch 128 on 8 GPUs -> per-GPU shard 16", dict(N=2048, W=48, H=12, multi=5, B=16)),
]
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (2 495 measured)
# HBM bytes / MFMA utilisation per family from the committed PMC passes, one file per arithmetic of the GLU products
TRAFFIC_FILES = {"bf16x2": os.path.join(ROOT, "profiles", "r06_pmc_traffic_bf16x2.json")}
TRAFFIC_FILE_F32 = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: become N ranks through torch.distributed.run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


# ---------------------------------------------------------------------------------------------------------------
# roofline: the MFMA GEMM families of the spectral GLU stack (94 % of the hot path's FLOPs)
def glu_flops(cfg):
    """(algorithmic, executed) FLOPs of ONE pass over the three GLU layers of one block, both branches.
    algorithmic: dense count of SURVEY 8d, 2*M*(C0*2C + 2C*C + 2C*C) per branch, C0 = 4W, C = 4*W*multi.
    executed: what the kernels multiply -- the zero k=0 Chebyshev slice is skipped (K 4W -> 3W), the dead C2R bins of
    the last layer are dropped (N 2C -> 2*ceil16(4*nf)), widths padded to multiples of 16."""
    B, N, W, multi = cfg["B"], cfg["N"], cfg["W"], cfg["multi"]
    M, Wm = B * N, W * multi
    C0, C = 4 * W, 4 * Wm

    def c16(v):
        return (v + 15) // 16 * 16
    alg = 2 * (2.0 * M * (C0 * 2 * C + C * 2 * C + C * 2 * C))
    CP = c16(C)
    cp2 = [c16(4 * (Wm // 2 + 1)), c16(max(4 * ((Wm + 1) // 2 - 1), 1))]
    exe = sum(2.0 * M * (3 * W * 2 * CP + CP * 2 * CP + CP * 2 * cp2[r]) for r in range(2))
    return alg, exe


def step_flops(cfg):
    """Algorithmic FLOPs of one train step of the hot path, SURVEY 8d: F_step = 3 F_fwd (GRU excluded, as there), and the
    GRU's own (forward: input projection + recurrent product; backward: the transposed recurrent product + dW_hh + dW_ih)."""
    B, N, W, multi = cfg["B"], cfg["N"], cfg["W"], cfg["multi"]
    M, Wm = B * N, W * multi
    C0, C = 4 * W, 4 * Wm
    f_fwd = 2.0 * (2 * 3 * N * N * B * W + 8 * M * C0 * C + 16 * M * C * C + 8 * M * Wm * Wm + 2 * M * Wm * Wm + 2 * M * Wm * W) \
        + 2.0 * M * Wm * W + 2.0 * M * W * W + 4.0 * N ** 3 + 12.0 * B * N * N
    rec = 2.0 * B * 3 * N * N * N                      # N steps of [B x N] . [N x 3N]
    proj = 2.0 * B * N * W * 3 * N
    return {"hot_path": 3.0 * f_fwd, "gru_fwd": rec + proj, "gru_bwd": rec + rec + proj}   # bwd: dh W_hh, dW_hh (same size), dW_ih


def time_gemm_families(cfg, iters=20):
    """Live HIP-event timing (on the launch stream) of the three GEMM families, through the C ABI on random operands.
    Returns {family: dict(us_per_call, launches_per_call, calls_per_step, kernel)}."""
    import torch
    from stemgnn_amd import _lib, ops

    lib = _lib.load()
    B, N, W, multi = cfg["B"], cfg["N"], cfg["W"], cfg["multi"]
    dev = torch.device("cuda")
    nsplit = ops._NSPLIT
    packed = torch.randn(lib.stemgnn_packed_floats(W, multi), device=dev) * 0.05
    saved = torch.randn(lib.stemgnn_saved_floats(B, N, W, multi), device=dev)
    scratch = torch.randn(lib.stemgnn_scratch_floats(B, N, W, multi), device=dev) * 0.1
    gradpart = torch.empty(lib.stemgnn_gradpart_floats(W, multi, nsplit), device=dev)
    st = torch.cuda.current_stream()

    def fwd():
        _lib.check(lib.stemgnn_spectral_glu_fwd(packed.data_ptr(), saved.data_ptr(), B, N, W, multi, st.cuda_stream), "glu_fwd")

    # STEMGNN_DTYPE=bf16x2 with 4 W multi <= 256: the forward and the data-gradient chain run on the bf16 matrix pipe
    # (csrc/glu_fused_bf16.h); the weight gradients stay exact fp32
    bf16 = ops.glu_splits() == 2 and bool(lib.stemgnn_glu_fused_bf16_ok(W, multi, 2))
    if bf16:
        split = torch.empty(lib.stemgnn_glu_split_floats(W, multi, 2), device=dev)
        _lib.check(lib.stemgnn_glu_fused_repack(packed.data_ptr(), W, multi, st.cuda_stream), "repack")
        _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st.cuda_stream), "split_panels")

    def fwd_bf16():
        _lib.check(lib.stemgnn_spectral_glu_fwd_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), B, N, W, multi, 2,
                                                      st.cuda_stream), "glu_fwd bf16x2")

    def dgrad_bf16():
        _lib.check(lib.stemgnn_spectral_glu_dgrad_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                                        B, N, W, multi, 2, st.cuda_stream), "glu_dgrad bf16x2")

    def bwd(parts):
        def run():
            _lib.check(lib.stemgnn_spectral_glu_bwd(packed.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                                    gradpart.data_ptr(), nsplit, parts, B, N, W, multi, st.cuda_stream),
                       "glu_bwd")
        return run

    # round 4: one fused launch per block for the forward and for the data-gradient chain where the padded channel count
    # 4 W multi is <= 256 (csrc/glu_fused.h); configs[4] (W = 48) keeps the per-layer launches
    fused = (4 * W * multi + 15) // 16 * 16 <= 256 and os.environ.get("STEMGNN_GLU_FUSED", "1") != "0"
    if fused:
        fwd_desc = (fwd, 1, "sg_glu_fused_fwd_kernel (the three GLU layers of a block, both branches, in ONE launch: row block's "
                            "activations resident in LDS, weights on a direct-to-LDS ring)")
        dg_desc = (bwd(1), 1, "sg_glu_fused_dgrad_kernel (d(pre-activation) of layer 2 -> 1 -> 0 -> dG in ONE launch per block)")
    else:
        fwd_desc = (fwd, 3, "sg_gemm2<GluFwdEpi> (spectral GLU forward: 3 launches per block, both branches per launch)")
        dg_desc = (bwd(1), 3, "sg_gemm2<GluDpreEpi> x2 + sg_gemm_f32<GluDgrad0Op> (GLU data gradients: 3 launches per block)")
    if bf16:
        fwd_desc = (fwd_bf16, 1, "sg_glu_fused_fwd_bf16_kernel (the three GLU layers of a block in ONE launch, split-bf16 products a_hi b_hi "
                                 "+ a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_bf16, activations as two bf16 planes in LDS)")
        dg_desc = (dgrad_bf16, 1, "sg_glu_fused_dgrad_bf16_kernel (d(pre-activation) of layer 2 -> 1 -> 0 -> dG in ONE launch, split-bf16 "
                                  "products on v_mfma_f32_32x32x16_bf16)")
    fams = {
        "glu_fwd": fwd_desc,
        "glu_dgrad": dg_desc,
        "glu_wgrad": (bwd(2), 1, "sg_wgrad_kernel (all six GLU weight-gradient products of a block in ONE launch: direct-to-LDS "
                                 "ring, in-kernel fixed-order split reduction -- no reduce kernel)"),
    }
    wg_bf16 = bf16 and os.environ.get("STEMGNN_WGRAD_BF16", "1") != "0"
    if wg_bf16:     # round 6: the weight gradients on the bf16 matrix pipe too (csrc/wgrad.h wg_stage_bf16)
        fams["glu_wgrad"] = (bwd(6), 1, "sg_wgrad_kernel<split-bf16> (all six GLU weight-gradient products of a block in ONE launch; fp32 "
                                        "operands split into bf16 hi / lo on the fly, a_hi b_hi + a_hi b_lo + a_lo b_hi on "
                                        "v_mfma_f32_32x32x16_bf16, same ring and fixed-order split reduction)")
    out = {}
    for name, (fn, launches, kernel) in fams.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        out[name] = dict(us_per_call=e0.elapsed_time(e1) * 1e3 / iters, launches_per_call=launches, calls_per_step=2,
                         kernel=kernel, bf16=bool(bf16 and (name != "glu_wgrad" or wg_bf16)))
    return out


def time_gru(cfg, iters=5):
    """The GRU front alone (ops.GruFront through the C ABI, HIP events on the launch stream): the recurrence is a chain of
    N dependent steps, so it is bound by the per-step exchange latency, not by FLOPs or bytes -- reported as microseconds per
    recurrence step, the figure that says what bounds the large configurations (reference models/base_model.py:137)."""
    import torch
    from stemgnn_amd import ops

    B, N, W = cfg["B"], cfg["N"], cfg["W"]
    dev = torch.device("cuda")
    gru = torch.nn.GRU(W, N).to(dev)
    ps = [gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0]
    x = torch.randn(B, W, N, device=dev)
    dh = torch.randn(N, B, N, device=dev) * 0.01

    def fwd_bwd():
        for p in ps:
            p.grad = None
        ops.GruFront.apply(x, *ps).backward(dh)
    fwd_bwd()
    st = torch.cuda.current_stream()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.no_grad():
        e[0].record(st)
        for _ in range(iters):
            ops.GruFront.apply(x, *ps)
        e[1].record(st)
    for _ in range(iters):
        fwd_bwd()
    e[2].record(st)
    e[2].synchronize()
    ops.check_gru_status(dev)
    f = e[0].elapsed_time(e[1]) * 1e3 / iters
    fb = e[1].elapsed_time(e[2]) * 1e3 / iters
    return {"bound": "latency (N dependent recurrence steps, one cross-workgroup exchange each)", "recurrence_steps": N,
            "fwd_us": f, "bwd_incl_weight_grads_us": fb - f, "fwd_us_per_recurrence_step": f / N,
            "bwd_us_per_recurrence_step": (fb - f) / N,
            "note": "fwd includes the input-projection GEMM, bwd the dW_hh / dW_ih weight-gradient launches; the recurrence "
                    "kernels alone are in profiles/r04_gru_wide_kernel_stats.txt (rocprofv3 of tools/gru_wide_time.py)"}


def step_roofline(cfg, ms_per_step):
    """The whole step against the fp32 matrix peak: SURVEY 8d's F_step (hot path, GRU excluded as there) over the step time."""
    f = step_flops(cfg)["hot_path"]
    tf = f / (ms_per_step * 1e-3) / 1e12
    return {"flops": f, "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
            "note": "algorithmic FLOPs of SURVEY 8d (F_step = 3 F_fwd, GRU excluded) / measured step time / fp32 MFMA peak"}


def rocprof_static_us():
    """Average launch durations (us) of the step's big kernels from the committed rocprofv3 --kernel-trace --stats summary of the
    ISOLATED timing loops (profiles/r06_kernel_stats_isolated*.txt, tools/gpu_job.sh prof_iso): constants, labelled as such --
    the live HIP-event figures beside them must agree."""
    name = "r06_kernel_stats_isolated_bf16x2.txt" if os.environ.get("STEMGNN_DTYPE") == "bf16x2" else "r06_kernel_stats_isolated.txt"
    path = os.path.join(ROOT, "profiles", name)
    out = {"source": "profiles/" + name}
    if not os.path.isfile(path):
        return out
    for ln in open(path):
        parts = ln.split()
        if len(parts) < 7 or parts[1] != "us/step":
            continue
        kernel = " ".join(parts[6:])
        for key in ("gru_fwd_cluster4_kernel", "gru_bwd_cluster4_kernel", "sg_glu_fused_fwd", "sg_glu_fused_dgrad",
                    "sg_wgrad_kernel<16, 6, false, true>", "sg_wgrad_kernel<16, 6, false, false>", "GruGiOp", "gru_gi_stream_kernel"):
            if key in kernel and key not in out:
                out[key] = float(parts[2])
    return out


def roofline_objects(cfg):
    fams = time_gemm_families(cfg)
    alg, exe = glu_flops(cfg)
    traffic = {}
    traffic_file = TRAFFIC_FILES.get(os.environ.get("STEMGNN_DTYPE", "f32"), TRAFFIC_FILE_F32)
    if os.path.isfile(traffic_file):
        with open(traffic_file) as f:
            traffic = json.load(f)
    rows = {}
    for name, t in fams.items():
        s = t["us_per_call"] * 1e-6                      # one call = the family's work for ONE block (3 layers x 2 branches)
        launches = t["launches_per_call"]
        # a split-bf16 family executes THREE bf16 products per fp32 product and is priced against the bf16 matrix peak; the
        # kernels are bound by their epilogues and HBM stores, not by the matrix pipe (profiles/r05_glu_fused_bf16_ablation.txt)
        peak = BF16_MFMA_PEAK_TFLOPS if t.get("bf16") else FP32_MFMA_PEAK_TFLOPS
        exe_f = 3.0 * exe if t.get("bf16") else exe
        rows[name] = {
            "kernel": t["kernel"], "bound": "mfma", "achieved": alg / s / 1e12, "achieved_executed": exe_f / s / 1e12,
            "peak": peak, "unit": "TFLOP/s", "frac": alg / s / 1e12 / peak,
            "frac_executed": exe_f / s / 1e12 / peak, "avg_launch_us": t["us_per_call"] / launches,
            "vs_fp32_mfma_peak": alg / s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "launches_per_step": launches * t["calls_per_step"], "sum_us_per_step": t["us_per_call"] * t["calls_per_step"],
            "flops_algorithmic": alg / launches, "flops_executed": exe / launches,
            # NOT measured in this run: constants from the committed rocprofv3 PMC passes (separate --pmc runs of this
            # bench, tools/pmc_summary.py), HBM bytes per launch / MFMA-busy fraction averaged over the family's launches
            "traffic": None,
            "traffic_static_pmc": traffic.get("kernels", {}).get(name),
            "mfma_util_static_pmc": traffic.get("mfma_util", {}).get(name),
            "traffic_static_pmc_source": traffic.get("source"),
        }
    # the GRU front: latency-bound recurrences, ALL of their time on the step's critical chain (the GLU weight gradients run
    # beside the chain on the side branch) -- priced against the same fp32 matrix peak so that the fractions are comparable
    gru = time_gru(cfg)
    fl = step_flops(cfg)
    for name, us, launches, kernel in (
            ("gru_fwd", gru["fwd_us"], 2, "gru_gi_stream_kernel / sg_gemm<GruGiOp> (input projection) + gru_fwd_cluster4_kernel / gru_fwd_wide_kernel "
                                          "(persistent recurrence, W_hh resident in registers, one exchange per step)"),
            ("gru_bwd", gru["bwd_incl_weight_grads_us"], 3, "zero-fill kernel + gru_bwd_cluster4_kernel / gru_bwd_wide_kernel "
                                                            "(persistent recurrence) + sg_wgrad_kernel (dW_hh, dW_ih sum)")):
        s = us * 1e-6
        rows[name] = {"kernel": kernel, "bound": "mfma", "limited_by": "latency: N dependent recurrence steps, one cross-workgroup "
                      "exchange each (DESIGN section 4)", "achieved": fl[name] / s / 1e12, "achieved_executed": fl[name] / s / 1e12,
                      "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl[name] / s / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                      "frac_executed": fl[name] / s / 1e12 / FP32_MFMA_PEAK_TFLOPS, "avg_launch_us": us / launches,
                      "vs_fp32_mfma_peak": fl[name] / s / 1e12 / FP32_MFMA_PEAK_TFLOPS, "launches_per_step": launches,
                      "sum_us_per_step": us, "flops_algorithmic": fl[name], "flops_executed": fl[name], "traffic": None,
                      "traffic_static_pmc": traffic.get("kernels", {}).get(name),       # the recurrence kernel alone
                      "mfma_util_static_pmc": traffic.get("mfma_util", {}).get(name),
                      "traffic_static_pmc_source": traffic.get("source"),
                      "us_per_recurrence_step": us / cfg["N"]}
    # critical-path time per step: the GLU forward and data-gradient launches and every GRU launch sit on the chain; the GLU
    # weight-gradient launches run on the side branch beside the backward chain / under the GRU recurrence
    for name, r in rows.items():
        r["on_critical_path"] = name != "glu_wgrad"
        r["critical_us_per_step"] = r["sum_us_per_step"] if r["on_critical_path"] else 0.0
    dominant = max(rows, key=lambda k: rows[k]["critical_us_per_step"])
    main = dict(rows[dominant])
    main["family"] = dominant
    if cfg == WORKLOAD:
        main["rocprof_isolated_avg_us_static"] = rocprof_static_us()
    main["why"] = ("the kernel family with the largest time on the step's CRITICAL PATH (each family timed live in isolation through "
                   "the C ABI, HIP events on the launch stream; frac = algorithmic FLOPs / time / fp32 MFMA peak); the MFMA GEMM "
                   "families are in roofline_families, the largest of them by summed GPU time in `roofline_mfma`")
    mf = max((k for k in rows if k.startswith("glu_")), key=lambda k: rows[k]["sum_us_per_step"])
    main_mfma = dict(rows[mf])
    main_mfma["family"] = mf
    return main, rows, main_mfma, gru


# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, budget_s=20.0):
    """The same train step on this box's host cores: the real reference (models/base_model.py Model + torch RMSprop,
    through oracle/ref_shim.py) when /root/reference is mounted, else the oracle port -- and the reason is stated.

    Thread count: the step is many small ops (a 228-step GRU, 228x228 softmax ...); on a many-core host using every
    core is pathologically slow (fork/join per tiny op), so a short calibration picks the fastest of {8, 16, 32, 64}
    threads (<= cpu_count) and `cores` reports what was actually used."""
    import torch
    from oracle import ref_shim

    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(cfg["B"], cfg["W"], cfg["N"], generator=g)
    y = torch.randn(cfg["B"], cfg["H"], cfg["N"], generator=g)
    extra = {}
    if ref_shim.reference_available():
        ref = ref_shim.load_reference_model_module()
        torch.manual_seed(0)
        model = ref.Model(cfg["N"], 2, cfg["W"], cfg["multi"], horizon=cfg["H"])      # dropout 0.5 as the driver leaves it
        model.train()
        opt = torch.optim.RMSprop(model.parameters(), lr=1e-4, eps=1e-8)
        lossf = torch.nn.MSELoss(reduction="mean")

        def one():
            model.zero_grad()
            forecast, _ = model(x)
            loss = lossf(forecast, y)
            loss.backward()
            opt.step()
        kind, what = "reference", "the reference's Model + torch.optim.RMSprop (models/handler.py:157-165 loop body)"
    else:
        from oracle.stemgnn_oracle import OracleTrainer
        tr = OracleTrainer(cfg["N"], cfg["W"], cfg["multi"], cfg["H"], lr=1e-4, seed=0, dropout_rate=0.5)

        def one():
            tr.step(x, y)
        kind, what = "port", "the torch-CPU oracle port"
        extra["why"] = (f"{ref_shim.REFERENCE_ROOT} is not mounted on this box (it exists only in the build container), "
                        "so the CPU restatement of the same step is timed instead")
    best_t, best_n = None, None
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one()                                            # warm-up at this thread count
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
        if dt > 5.0:
            break
    torch.set_num_threads(best_n)
    one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 100:
            break
    sps = n / el
    out = {"value": cfg["B"] * cfg["H"] * sps, "unit": "forecast-steps/s", "cores": best_n, "kind": kind,
           "ms_per_step": 1e3 / sps, "host_cpus": ncpu,
           "sample": f"{n} train steps of {what} (same shape, batch {cfg['B']}, fp32, RMSprop, dropout 0.5), "
                     f"{best_n} threads (fastest of a 8/16/32/64 calibration) on a {ncpu}-cpu host"}
    # the real reference's figure for this shape, measured in the BUILD container (8-core Xeon, torch
    # 2.10 CPU, 0.2918 s per step) -- not interchangeable with a "port" number from this box's host, shown beside it
    if cfg == WORKLOAD:
        out["reference_container_ms"] = 291.8
        out["reference_container_note"] = "BASELINE.md section 3: the reference itself, 8-core build container, other host"
    out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------------------------
def run_training(cfg, steps, warmup, dev, world, rank, graph=True, T=12672, collective=None, extra=None, short=0):
    """Build model + resident series, run warmup + `steps` timed train steps; returns (elapsed_s, mode, final_loss).
    extra: a dict that receives `schedule` (engine.TrainStep's start-up self-check) and, with short > 0, `short_ms_per_step`:
    a second timed region of `short` steps right behind the first (the driver's command line is --steps 20)."""
    import torch
    import torch.distributed as dist
    from stemgnn_amd import Model, ops
    from stemgnn_amd.distributed import broadcast_parameters
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop

    torch.manual_seed(0)
    model = Model(cfg["N"], 2, cfg["W"], cfg["multi"], horizon=cfg["H"])      # defaults: dropout 0.5, leaky 0.2
    model.to(dev).train()
    broadcast_parameters(model)
    # same arithmetic as the driver's torch.optim.RMSprop (handler.py:127), one fused kernel (+ grad zeroing)
    opt = FusedRMSprop(model.parameters(), lr=1e-4, alpha=0.99, eps=1e-8)
    # synthetic z-scored series resident in HBM (PEMS07 length), windows gathered by index inside the step
    g = torch.Generator().manual_seed(1234 + rank)
    series = torch.randn(T, cfg["N"], generator=g).to(dev)
    n_windows = T - cfg["W"] - cfg["H"] + 1
    total = steps + warmup + 1 + short
    epochs = -(-total * cfg["B"] // n_windows)                       # shuffled passes over the windows, back to back
    order = torch.cat([torch.randperm(n_windows, generator=g) for _ in range(epochs)])[: total * cfg["B"]]
    hi_all = (order + cfg["W"]).to(dev).view(total, cfg["B"])        # window-end rows (ForecastDataset.x_end_idx)

    # the shuffled order is loaded once (device-side iterator, stemgnn_window_gather_queue): per step the host only
    # replays the graph; every step still gathers its own B fresh windows from the series inside the timed region
    stepper = TrainStep(model, opt, cfg["B"], cfg["W"], cfg["H"], cfg["N"], series=series, world=world, graph=graph,
                        collective=collective, order_capacity=total * cfg["B"])
    stepper.load_order(hi_all)
    stepper.run_next()                      # eager step (lazy init of tables, seed, state) + graph capture
    torch.cuda.synchronize()
    for _ in range(warmup):
        stepper.run_next()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        stepper.run_next()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if short:
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(short):
            stepper.run_next()
        torch.cuda.synchronize()
        if extra is not None:
            extra["short_ms_per_step"] = (time.perf_counter() - t1) / short * 1e3
    if extra is not None:
        extra["schedule"] = dict(stepper.schedule)
    if dist.is_initialized() and extra is not None and stepper.bucket is not None:
        # the data-parallel step's collectives alone, eager, on every rank together (outside the timed region; the gradient
        # buffer holds zeros after the fused optimizer step): the head range [0, split) is the one all-reduce the step EXPOSES
        # (behind the GRU weight gradients), the tail range rides on the side branch under the GRU recurrence
        flat, split = stepper.bucket.flat, stepper._split
        ar = {"two_range": split is not None, "flat_bytes": flat.numel() * 4}

        def timed(view, n=20):
            for _ in range(3):
                dist.all_reduce(view)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                dist.all_reduce(view)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        ar["flat_allreduce_us"] = timed(flat)
        if split is not None:
            ar.update(head_bytes=split * 4, tail_bytes=(flat.numel() - split) * 4,
                      exposed_head_allreduce_us=timed(flat[:split]), tail_allreduce_us=timed(flat[split:]))
        extra["allreduce"] = ar
    ops.check_gru_status(dev)               # outside the timed region: a lost GRU cluster partner must fail the run
    ops.check_gather_status(dev)
    final_loss = float(stepper.loss.item())
    if not (final_loss == final_loss) or final_loss > 1e6:
        raise SystemExit(f"training diverged: loss={final_loss}")
    return elapsed, stepper.mode, final_loss


def workload_name(cfg):
    return (("PEMS07-shape " if cfg == WORKLOAD else "") +
            f"N={cfg['N']} W={cfg['W']} H={cfg['H']} multi={cfg['multi']} stack=2, batch {cfg['B']} per GPU, "
            "train step (window gather+fwd+MSE+bwd+RMSprop), dropout 0.5, arithmetic " + os.environ.get("STEMGNN_DTYPE", "f32"))


# the reference itself on the 8-core build container (s per step), per BASELINE config -- another host,
# shown beside the live figure of this box and labelled as such
REFERENCE_CONTAINER_S = {
    "configs[0]": (0.1365, "ECG shape N=140 B=32"), "configs[2]": (0.5499, "PEMS03 shape N=358 B=32"),
    "configs[3]": (11.50, "N=1024 at the GLOBAL batch 64 (the per-GPU shard of the row is 8)"),
    "configs[4]": (30.8, "N=2048 W=48 H=12 at the per-GPU shard batch 16"),
}


def cpu_steps_brief(cfg, max_s=12.0):
    """1 warm-up + up to 2 timed train steps of the oracle port (or the reference when mounted) on this box's host cores;
    None when one step does not finish inside `max_s` (the labelled container number stands alone then)."""
    import torch
    from oracle.stemgnn_oracle import OracleTrainer

    g = torch.Generator().manual_seed(0)
    x = torch.randn(cfg["B"], cfg["W"], cfg["N"], generator=g)
    y = torch.randn(cfg["B"], cfg["H"], cfg["N"], generator=g)
    nt = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nt)
    tr = OracleTrainer(cfg["N"], cfg["W"], cfg["multi"], cfg["H"], lr=1e-4, seed=0, dropout_rate=0.5)
    t0 = time.perf_counter()
    tr.step(x, y)
    first = time.perf_counter() - t0
    if first > max_s:
        return {"kind": "port", "cores": nt, "ms_per_step": first * 1e3, "steps": 1,
                "sample": "the single (cold) step of the torch-CPU oracle port; no second step inside the time budget"}
    n, t0 = 0, time.perf_counter()
    while n < 2 and time.perf_counter() - t0 < max_s:
        tr.step(x, y)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"kind": "port", "cores": nt, "ms_per_step": dt * 1e3, "value": cfg["B"] * cfg["H"] / dt, "unit": "forecast-steps/s",
            "steps": n, "sample": f"{n} train steps of the torch-CPU oracle port after one warm-up, {nt} threads"}


def section_other_configs(args, dev):
    import torch
    others = []
    for name, c in OTHER_CONFIGS:
        try:
            torch.cuda.empty_cache()
            big = c["N"] >= 1024
            k, w = (5, 2) if big else (30, 5)
            el, md, _ = run_training(c, k, w, dev, 1, 0, graph=not args.no_graph, T=4096 if big else 12672)
            row = {"config": name, "workload": workload_name(c), "ms_per_step": el / k * 1e3,
                   "value": c["B"] * c["H"] / (el / k), "unit": "forecast-steps/s", "steps": k,
                   "warmup": w, "n_gpus": 1, "launch": md}
            if not args.no_roofline:        # what bounds this shape: the GEMM families' fractions + the GRU's latency floor
                torch.cuda.empty_cache()
                _, fams, _, gru = roofline_objects(c)
                row["roofline_families"] = {
                    f: {q: v[q] for q in ("frac", "frac_executed", "achieved", "avg_launch_us", "sum_us_per_step",
                                          "launches_per_step", "on_critical_path")} for f, v in fams.items()}
                row["gru"] = gru
                row["gru"]["critical_us"] = row["gru"]["fwd_us"] + row["gru"]["bwd_incl_weight_grads_us"]
                row["gru"]["share_of_step"] = row["gru"]["critical_us"] / (row["ms_per_step"] * 1e3)
                row["glu_gemm_share_of_step"] = sum(v["sum_us_per_step"] for f, v in fams.items()
                                                    if f.startswith("glu_")) / (row["ms_per_step"] * 1e3)
                row["glu_kernels"] = ("fused (one launch per block)" if (4 * c["W"] * c["multi"] + 15) // 16 * 16 <= 256
                                      else "per-layer launches (4 W multi > 256: the row block's activations do not fit the LDS)")
                row["step_roofline"] = step_roofline(c, row["ms_per_step"])
            key = name.split(" ")[0]
            cpu = {"reference_container_s_per_step": REFERENCE_CONTAINER_S[key][0],
                   "reference_container_note": "BASELINE.md section 3 (the reference itself, 8-core build container, another "
                                               "host): " + REFERENCE_CONTAINER_S[key][1]}
            if not args.no_cpu_baseline:      # configs[4]: ~30 s per step -> its ONE (cold) step is the figure, labelled as such
                try:
                    cpu.update(cpu_steps_brief(c))
                except Exception as e:  # noqa: BLE001
                    cpu["error"] = f"{type(e).__name__}: {e}"
            row["cpu_baseline"] = cpu
            others.append(row)
        except Exception as e:  # noqa: BLE001 -- a failing side line must not lose the others
            others.append({"config": name, "error": f"{type(e).__name__}: {e}"})
    return others


def section_dtype_variants(args, dev, cfg):
    """BASELINE.json configs[1] names "bf16/fp32": the headline is exact fp32 (the reference's arithmetic); the same step with
    the GLU forward / data-gradient layers as split-bf16 products (STEMGNN_DTYPE, csrc/gemm2s.h) is reported beside it,
    with the model-level error each setting was tested to (tests/test_hip_splitgemm.py)."""
    import torch
    variants = []
    before = os.environ.get("STEMGNN_DTYPE")
    for dt, err in (("bf16x2", "<= 3e-5 norm-relative vs the oracle (gate: 1e-4)"),
                    ("f32", "<= 9e-6 norm-relative vs the oracle (exact fp32 products; fp32 re-association only)"),
                    ("bf16x3", "<= 4e-6 norm-relative vs the oracle (fp32 class)")):
        if dt == (before or "f32"):
            continue                      # the headline's own arithmetic
        os.environ["STEMGNN_DTYPE"] = dt
        try:
            torch.cuda.empty_cache()
            el, md, _ = run_training(cfg, 60, 10, dev, 1, 0, graph=not args.no_graph)
            row = {"dtype": dt, "ms_per_step": el / 60 * 1e3, "value": cfg["B"] * cfg["H"] / (el / 60),
                   "unit": "forecast-steps/s", "steps": 60, "warmup": 10, "launch": md,
                   "arithmetic": DTYPE_NOTE[dt], "tested_error": err}
            if not args.no_roofline and dt != "bf16x3":
                torch.cuda.empty_cache()
                row["roofline"], fams, row["roofline_mfma"], _ = roofline_objects(cfg)
                row["roofline_families"] = {
                    f: {q: v[q] for q in ("kernel", "frac", "frac_executed", "achieved", "peak", "avg_launch_us", "sum_us_per_step",
                                          "launches_per_step", "on_critical_path", "traffic", "traffic_static_pmc",
                                          "mfma_util_static_pmc") if q in v} for f, v in fams.items()}
                row["step_roofline"] = step_roofline(cfg, row["ms_per_step"])
            variants.append(row)
        except Exception as e:  # noqa: BLE001
            variants.append({"dtype": dt, "error": f"{type(e).__name__}: {e}"})
        finally:
            if before is None:
                os.environ.pop("STEMGNN_DTYPE", None)
            else:
                os.environ["STEMGNN_DTYPE"] = before
    return variants


def time_eigh(N, iters=10, batch=1):
    """The eigensolver stage alone (stemgnn_eigh_fwd through the C ABI, HIP events on the launch stream) on a Laplacian-like
    symmetric matrix: microseconds per decomposition (north_star's first component; reference: none, nearest code
    models/base_model.py:121-134)."""
    import torch
    from stemgnn_amd import _lib, ops

    lib = _lib.load()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(N)
    A = torch.rand(N, N, generator=g)
    A = 0.5 * (A + A.t())
    d = A.sum(1)
    L = (torch.diag(d) - A) / torch.sqrt(d[:, None] * d[None, :])
    mul_L = torch.zeros(4, N, N)
    mul_L[1] = L
    mul_L = mul_L.to(dev)
    lam = torch.empty(N, device=dev)
    U = torch.empty(N, N, device=dev)
    scr = torch.empty(lib.stemgnn_eigh_scratch_floats(N), device=dev)
    st = torch.cuda.current_stream()

    def run():
        _lib.check(lib.stemgnn_eigh_fwd(mul_L.data_ptr(), lam.data_ptr(), U.data_ptr(), scr.data_ptr(), N, 0, st.cuda_stream), "eigh_fwd")
    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    one = timed(run)
    ops.check_eigh_status(dev)
    out = {"eigh_us": one}
    if batch > 1:       # north_star's "batched" eigensolver: `batch` Laplacians in one call (stemgnn_eigh_batched)
        mb = mul_L.unsqueeze(0).repeat(batch, 1, 1, 1).contiguous()
        lamb, Ub = torch.empty(batch, N, device=dev), torch.empty(batch, N, N, device=dev)
        scrb = torch.empty(batch * lib.stemgnn_eigh_scratch_floats(N), device=dev)

        def run_b():
            _lib.check(lib.stemgnn_eigh_batched(mb.data_ptr(), lamb.data_ptr(), Ub.data_ptr(), scrb.data_ptr(), N, batch,
                                                st.cuda_stream), "eigh_batched")
        tb = timed(run_b)
        ops.check_eigh_status(dev)
        out.update(eigh_batch=batch, eigh_batched_us=tb, eigh_batched_us_per_matrix=tb / batch)
    return out


def section_spectral_variants(args, dev, cfg):
    """north_star's eigen route (STEMGNN_SPECTRAL=eig: T_k(L) = U p_k(Lambda) U^T from stemgnn_eigh_fwd instead of the two
    Chebyshev products): the headline step and the configs[3] shard with it, and the solver's own time."""
    import torch
    rows = []
    before = os.environ.get("STEMGNN_SPECTRAL")
    os.environ["STEMGNN_SPECTRAL"] = "eig"
    try:
        for name, c, k, w, T in (("configs[1] PEMS07 shape", cfg, 60, 10, 12672),
                                 ("configs[3] N=1024 shard (batch 8)", OTHER_CONFIGS[2][1], 5, 2, 4096)):
            try:
                torch.cuda.empty_cache()
                el, md, _ = run_training(c, k, w, dev, 1, 0, graph=not args.no_graph, T=T)
                rows.append({"config": name, "spectral": "eig", "ms_per_step": el / k * 1e3,
                             "value": c["B"] * c["H"] / (el / k), "unit": "forecast-steps/s", "steps": k, "warmup": w,
                             "launch": md, "N": c["N"], **time_eigh(c["N"], batch=8 if c["N"] <= 256 else 1)})
            except Exception as e:  # noqa: BLE001
                rows.append({"config": name, "spectral": "eig", "error": f"{type(e).__name__}: {e}"})
    finally:
        if before is None:
            os.environ.pop("STEMGNN_SPECTRAL", None)
        else:
            os.environ["STEMGNN_SPECTRAL"] = before
    return rows


def section_mae_vs_ref(args, dev):
    """The metric's second half, "MAE vs ref", in the bench line: the reference's own `handler.train` run at the headline
    shape (N=228, W=12, H=3, multi=5, batch 32; 2 epochs, dropout pinned to 0 so the run is replayable) is committed as
    tests/golden/data/train_pems07.npz (made by tests/golden/make_golden_data.py from the unmodified reference); the same
    series / seed / schedule is trained here through stemgnn_amd.trainer.train (hipGraph step) and the validation MAE after
    every epoch is set beside the reference's.  A CHECK against committed reference output (like the parity tests), not a
    timed leg; the series comes from the fixture's deterministic generator (tests/util.synthetic_series)."""
    import tempfile
    import types

    import numpy as np
    import torch
    from stemgnn_amd import Model, trainer
    from tests.util import synthetic_series

    z = np.load(os.path.join(ROOT, "tests", "golden", "data", "train_pems07.npz"))
    T, N, W, H, multi, bs, epochs, ntrain = (int(v) for v in z["cfg"])
    raw = synthetic_series(T, N, int(z["raw_seed"]))
    a = types.SimpleNamespace(window_size=W, horizon=H, multi_layer=multi, device=str(dev), norm_method="z_score",
                              optimizer="RMSProp", lr=float(z["lr"]), decay_rate=0.5, exponential_decay_step=2,
                              batch_size=bs, epoch=epochs, validate_freq=1, early_stop=False, hipgraph=not args.no_graph)
    vals, losses = [], []
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            trainer.train(raw[:ntrain], raw[ntrain:], a, tmp, model_factory=lambda *x, **k: Model(*x, dropout_rate=0.0, **k),
                          on_step=lambda e, i, st: losses.append(st.loss.clone()), on_validate=lambda e, m: vals.append(m))
    got_loss = torch.stack(losses).double().cpu().numpy()
    rows = []
    for e in range(epochs):
        ref = {k: float(z[f"val{e}_{k}"]) for k in ("mae", "mape", "rmse")}
        hip = {k: float(vals[e][k]) for k in ("mae", "mape", "rmse")}
        rows.append({"epoch": e, "mae_hip": hip["mae"], "mae_ref": ref["mae"], "mae_rel_diff": abs(hip["mae"] - ref["mae"]) / ref["mae"],
                     "rmse_hip": hip["rmse"], "rmse_ref": ref["rmse"], "mape_hip": hip["mape"], "mape_ref": ref["mape"]})
    return {"workload": f"reference handler.train at N={N} W={W} H={H} multi={multi} batch {bs}: {epochs} epochs over {ntrain} "
                        f"rows, validation on {T - ntrain} rows, RMSprop lr {float(z['lr'])}, dropout 0, seed 0",
            "reference_run": "tests/golden/data/train_pems07.npz (unmodified reference, CPU, committed fixture)",
            "epochs": rows, "train_loss_max_rel_diff": float(np.max(np.abs(got_loss - z["losses"]) / np.abs(z["losses"]))),
            "mae_vs_ref_max_rel_diff": max(r["mae_rel_diff"] for r in rows)}


SECTIONS = ("other_configs", "dtype_variants", "spectral_variants", "mae_vs_ref")


def run_section_child(name, args):
    """Run one side section in a child process (`bench.py --section name`): a GPU fault or abort there costs that section,
    not the headline.  Returns the child's JSON (or an error object)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--section", name]
    for flag, on in (("--no-graph", args.no_graph), ("--no-roofline", args.no_roofline), ("--no-cpu-baseline", args.no_cpu_baseline)):
        if on:
            cmd.append(flag)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "section timed out after 900 s"}
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") or ln.startswith("[")]
    if p.returncode != 0 or not lines:
        return {"error": f"child exited with {p.returncode}", "stderr_tail": p.stderr[-400:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip every side section (other_configs, dtype / spectral variants)")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the isolated GEMM-family timing loops (for a kernel trace that holds in-step launches only)")
    ap.add_argument("--section", choices=SECTIONS, help="internal: run ONE side section and print its JSON (child process)")
    ap.add_argument("--dtype", choices=tuple(DTYPE_NOTE), default=None,
                    help=f"arithmetic of the GLU products (default: $STEMGNN_DTYPE, else {DEFAULT_DTYPE}); f32 = the library default")
    args = ap.parse_args()
    dtype = args.dtype or os.environ.get("STEMGNN_DTYPE") or DEFAULT_DTYPE
    os.environ["STEMGNN_DTYPE"] = dtype       # read per call by stemgnn_amd.ops.glu_splits; inherited by the section children

    if args.section:
        import torch
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        cfg = bench_workload()
        res = {"other_configs": lambda: section_other_configs(args, dev),
               "dtype_variants": lambda: section_dtype_variants(args, dev, cfg),
               "spectral_variants": lambda: section_spectral_variants(args, dev, cfg),
               "mae_vs_ref": lambda: section_mae_vs_ref(args, dev)}[args.section]()
        print(json.dumps(res), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")          # non-zero exit code: the run is not what was asked
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # Under a launcher (RANK / WORLD_SIZE in the environment) the run is a data-parallel job even with ONE rank: the RCCL
    # process group is created and the step runs the collective form -- what the N > 1 runs execute, testable on a 1-GPU
    # box.  Plain `python bench.py` (the driver's N = 1 line) has no process group.
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # flight recorder on: engine.capture waits until the RCCL watchdog has retired the eager warm-up collectives by
        # reading the recorder (a drain on observed state); without it that wait is a fixed 0.35 s
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus or dist.get_backend() != "nccl":
            raise SystemExit(f"RCCL process group has {dist.get_world_size()} ranks over {dist.get_backend()!r}, --gpus asked for "
                             f"{args.gpus} over nccl (= RCCL)")

    cfg = bench_workload()
    extra = {}
    elapsed, mode, final_loss = run_training(cfg, args.steps, args.warmup, dev, world, rank, graph=not args.no_graph,
                                             collective=True if (world > 1 or launched) else None, extra=extra,
                                             short=20 if args.steps > 20 else 0)
    out = {
        "metric": "forecast-steps/sec (train)", "value": world * cfg["B"] * cfg["H"] / (elapsed / args.steps),
        "unit": "forecast-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "dtype_note": DTYPE_NOTE[dtype], "data": "synthetic",
        "config": {"workload": workload_name(cfg), "global_batch": world * cfg["B"], "per_gpu_batch": cfg["B"],
                   "parallelism": f"dp{world}", "launch": mode},
        "final_loss": final_loss,
        "steps_per_s": args.steps / elapsed, "samples_per_s": world * cfg["B"] * args.steps / elapsed,   # also reported
        # engine.TrainStep's start-up self-check of the hipGraph's two-branch schedule: the captured step
        # against the same step with the side branch serialised and against the side branch's own kernel-time sum
        "schedule": extra.get("schedule"),
    }
    if dist.is_initialized():
        # the first multi-GPU run explains itself: how many ranks RCCL saw, which form of the step was adopted, whether the
        # one-graph two-range schedule passed its start-up verification on the REAL collectives, and what the collectives cost
        sch = extra.get("schedule") or {}
        out["multi_gpu"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "adopted_mode": mode,
                            "one_graph_verified": sch.get("one_graph_verified"), "adopted_verified": sch.get("adopted_verified"),
                            "recapture_verified": sch.get("recapture_verified"), "tail_hook_missed": sch.get("tail_hook_missed", 0),
                            "capture_drain": sch.get("capture_drain"), "allreduce": extra.get("allreduce")}
    if "short_ms_per_step" in extra:        # what the driver's `--steps 20 --warmup 5` command measures, from the same process
        out["ms_per_step_20_steps"] = extra["short_ms_per_step"]
    if rank == 0:
        print("bench headline (complete line follows on stdout): " + json.dumps(out), file=sys.stderr, flush=True)
        if not args.no_roofline:
            out["roofline"], out["roofline_families"], out["roofline_mfma"], out["gru"] = roofline_objects(cfg)
            out["gru"]["critical_us"] = out["gru"]["fwd_us"] + out["gru"]["bwd_incl_weight_grads_us"]
            out["gru"]["share_of_step"] = out["gru"]["critical_us"] / (out["ms_per_step"] * 1e3)
            out["step_roofline"] = step_roofline(cfg, out["ms_per_step"])
        solo = world == 1 and not launched
        if solo and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        if solo and not args.no_other_configs:
            torch.cuda.empty_cache()
            for name in SECTIONS:
                out[name] = run_section_child(name, args)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()                      # rank 0 may still be timing the roofline kernels
        dist.destroy_process_group()


def bench_workload():
    cfg = dict(WORKLOAD)
    if os.environ.get("STEMGNN_BENCH_WORKLOAD"):     # "N,W,H,multi,B": another shape as the main line (experiments only)
        vals = [int(v) for v in os.environ["STEMGNN_BENCH_WORKLOAD"].split(",")]
        cfg = dict(zip(("N", "W", "H", "multi", "B"), vals))
    return cfg


if __name__ == "__main__":
    main()
