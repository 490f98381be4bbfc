"""Statistical "MAE vs ref" fixture with the reference's REAL dropout (0.5).

    python tests/golden/make_golden_dropout.py          (build container only: needs /root/reference)

The reference driver never forwards a dropout rate (models/handler.py:105), so every real run trains with
nn.Dropout(0.5) on the attention matrix (models/base_model.py:86,161).  Dropout masks come from the framework RNG, so a
bit-for-bit replay is impossible across implementations; what CAN be pinned is the distribution: this script runs the
UNMODIFIED reference `models.handler.train` for SEEDS different torch seeds on one synthetic series and commits, per
epoch, mean / std / min / max over the seeds of the validation MAE (raw and normalised units), MAPE, RMSE and of the mean
training loss -- plus the same run at dropout 0 as a sensitivity marker.  tests/test_hip_data.py then trains the HIP
path with its own Philox dropout for several seeds and must land inside that band.
Writes tests/golden/data/train_dropout_stats.npz.
"""
import contextlib
import io
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "data")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.ref_shim import load_reference_packages  # noqa: E402
from tests.util import synthetic_series  # noqa: E402

CFG = dict(T=420, N=32, W=12, H=3, multi=5, bs=32, epochs=6, ntrain=300, lr=1e-3, raw_seed=404)
SEEDS = list(range(12))


def run(hd, seed, dropout=None):
    c = CFG
    raw = synthetic_series(c["T"], c["N"], c["raw_seed"])
    args = types.SimpleNamespace(window_size=c["W"], horizon=c["H"], multi_layer=c["multi"], device="cpu",
                                 norm_method="z_score", optimizer="RMSProp", lr=c["lr"], decay_rate=0.5,
                                 exponential_decay_step=5, batch_size=c["bs"], epoch=c["epochs"], validate_freq=1,
                                 early_stop=False)
    vals = []
    real_validate, real_model = hd.validate, hd.Model

    def logged(*a, **k):
        r = real_validate(*a, **k)
        vals.append(r)
        return r
    hd.validate = logged
    if dropout is not None:
        hd.Model = lambda *a, **k: real_model(*a, dropout_rate=dropout, **k)
    torch.manual_seed(seed)                                       # main.py:52 seeds once, then builds the model
    buf = io.StringIO()
    try:
        with tempfile.TemporaryDirectory() as d, contextlib.redirect_stdout(buf):
            hd.train(raw[:c["ntrain"]], raw[c["ntrain"]:], args, d)
    finally:
        hd.validate, hd.Model = real_validate, real_model
    losses = [float(m) for m in re.findall(r"train_total_loss ([0-9.]+)", buf.getvalue())]
    norm_mae = [float(m) for m in re.findall(r"NORM: MAPE [0-9.%]+; MAE ([0-9.]+)", buf.getvalue())]
    return dict(mae=[float(v["mae"]) for v in vals], mape=[float(v["mape"]) for v in vals],
                rmse=[float(v["rmse"]) for v in vals], loss=losses, mae_norm=norm_mae)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    _, _, hd = load_reference_packages()
    runs = [run(hd, s) for s in SEEDS]
    p0 = run(hd, 0, dropout=0.0)
    out = dict(cfg=np.array([CFG[k] for k in ("T", "N", "W", "H", "multi", "bs", "epochs", "ntrain")], np.int64),
               lr=np.float64(CFG["lr"]), raw_seed=np.int64(CFG["raw_seed"]), seeds=np.asarray(SEEDS, np.int64))
    for k in ("mae", "mape", "rmse", "loss", "mae_norm"):
        arr = np.asarray([r[k] for r in runs], np.float64)            # [seed, epoch]
        out[k + "_runs"] = arr
        out[k + "_mean"], out[k + "_std"] = arr.mean(axis=0), arr.std(axis=0, ddof=1)
        out[k + "_p0"] = np.asarray(p0[k], np.float64)
        print(f"{k:9s} final: mean {arr[:, -1].mean():.5f}  std {arr[:, -1].std(ddof=1):.5f}  "
              f"min {arr[:, -1].min():.5f}  max {arr[:, -1].max():.5f}   dropout-0 run: {p0[k][-1]:.5f}")
    path = os.path.join(OUT, "train_dropout_stats.npz")
    np.savez_compressed(path, **out)
    print(f"train_dropout_stats: {os.path.getsize(path) / 1024:.1f} KiB")
