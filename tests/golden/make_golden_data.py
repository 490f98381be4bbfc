"""Generate golden vectors for the data path either side of the hot path from the REAL reference.

    python tests/golden/make_golden_data.py            (build container only: needs /root/reference)

Writes tests/golden/data/<case>.npz:
  norm_{z_score,min_max}  ForecastDataset(...).data (float64), x_end_idx, sample windows, de_normalized()
  metrics                 utils.math_utils.evaluate (all four axis variants, raw and normalised)
  rolling                 models.handler.inference with a stub model whose output is shorter than the horizon
  train_e2e               models.handler.train (3 epochs, RMSProp + ExponentialLR, validate every epoch) on a tiny
                          synthetic CSV with the reference Model at dropout 0: initial/final state_dict, batch order,
                          per-step losses, per-epoch validation metrics
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "data")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.detrand import det_normalish, det_uniform  # noqa: E402
from oracle.ref_shim import load_reference_packages  # noqa: E402
from tests.util import synthetic_series  # noqa: E402


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def case_norm(fd, method):
    T, N, W, H = 40, 6, 5, 3
    raw = synthetic_series(T, N, 101)
    raw[:, 2] = 3.25                                            # constant column: std == 0 -> 1, min_max scale = eps
    raw[0, 0] = np.nan; raw[1, 0] = np.nan                      # leading NaNs (bfill)
    raw[T - 1, 1] = np.nan                                      # trailing NaN (ffill)
    raw[17:20, 3] = np.nan                                      # interior run
    train = raw[:28]
    filled_train = fd.ForecastDataset(train, W, H).data         # fillna only (normalize_method=None)
    if method == "z_score":                                     # statistics as models/handler.py:112-119 makes them
        stat = {"mean": np.nanmean(train, axis=0).tolist(), "std": np.nanstd(train, axis=0).tolist()}
    else:
        stat = {"min": np.nanmin(train, axis=0).tolist(), "max": np.nanmax(train, axis=0).tolist()}
    out = dict(raw=raw, cfg=np.array([T, N, W, H], np.int64), filled_train=filled_train)
    for k, v in stat.items():
        out["stat_" + k] = np.asarray(v, np.float64)
    # quirk: with min_max the reference cannot take the *lists* handler.py:116-119 builds (`list - list` TypeError at
    # forecast_dataloader.py:11), so min_max only works with array statistics; z_score takes the lists.
    as_ref = (lambda v: list(v)) if method == "z_score" else (lambda v: np.asarray(v))
    for interval in (1, 3):
        ds = fd.ForecastDataset(raw, window_size=W, horizon=H, normalize_method=method,
                                norm_statistic={k: as_ref(v) for k, v in stat.items()}, interval=interval)
        out[f"x_end_idx_i{interval}"] = np.asarray(ds.x_end_idx, np.int64)
        if interval == 1:
            out["data"] = np.asarray(ds.data, np.float64)
            xs, ys = zip(*(ds[i] for i in range(len(ds))))
            out["x_all"] = torch.stack(xs).numpy()
            out["y_all"] = torch.stack(ys).numpy()
    own = fd.ForecastDataset(raw, window_size=W, horizon=H, normalize_method=method)   # statistics of the data itself
    out["data_ownstat"] = np.asarray(own.data, np.float64)
    f32 = det_normalish((7, H, N), 55)
    out["denorm_in"] = f32
    out["denorm_out"] = np.asarray(fd.de_normalized(f32, method, {k: as_ref(v) for k, v in stat.items()}), np.float64)
    out["denorm_out_f64in"] = np.asarray(
        fd.de_normalized(f32.astype(np.float64), method, {k: as_ref(v) for k, v in stat.items()}), np.float64)
    save("norm_" + method, **out)


def case_metrics(fd, mu):
    C, H, N = 37, 3, 5
    target = det_normalish((C, H, N), 71)
    target[3, 1, 2] = 0.0                                       # division by zero -> inf -> clipped to 5
    target[9, 0, 4] = 1e-4                                      # huge ratio -> clipped
    forecast = (target + 0.3 * det_normalish((C, H, N), 72)).astype(np.float32).astype(np.float64)
    stat = {"mean": det_uniform((N,), 73, -1.0, 4.0).astype(np.float64).tolist(),
            "std": det_uniform((N,), 74, 0.5, 2.0).astype(np.float64).tolist()}
    out = dict(target=target, forecast=forecast, stat_mean=np.asarray(stat["mean"]), stat_std=np.asarray(stat["std"]))
    t_raw = fd.de_normalized(target, "z_score", stat)
    f_raw = fd.de_normalized(forecast, "z_score", stat)
    for tag, (t, f) in dict(norm=(target, forecast), raw=(t_raw, f_raw)).items():
        for by_step in (False, True):
            for by_node in (False, True):
                m = mu.evaluate(t, f, by_step=by_step, by_node=by_node)
                for nm, v in zip(("mape", "mae", "rmse"), m):
                    out[f"{tag}_s{int(by_step)}n{int(by_node)}_{nm}"] = np.asarray(v, np.float64)
    save("metrics", **out)


class StubModel(torch.nn.Module):
    """forecast[b,j,n] = 0.5 * inputs[b, W-1-j, n] + 0.25 : every operation exact or singly rounded in fp32."""

    def __init__(self, L):
        super().__init__()
        self.L = L

    def forward(self, x):
        W = x.shape[1]
        return torch.stack([0.5 * x[:, W - 1 - j, :] + 0.25 for j in range(self.L)], dim=1), None


def case_rolling(fd, hd):
    T, N, W, horizon, L, bs = 30, 4, 6, 5, 2, 7
    raw = synthetic_series(T, N, 202)
    ds = fd.ForecastDataset(raw, window_size=W, horizon=horizon, normalize_method="z_score")
    loader = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0)
    f, t = hd.inference(StubModel(L), loader, "cpu", N, W, horizon)
    save("rolling", raw=raw, cfg=np.array([T, N, W, horizon, L, bs], np.int64), forecast=f, target=t)


def case_train(fd, hd, name="train_e2e", T=150, N=8, W=6, H=3, multi=2, bs=16, epochs=3, ntrain=110, lr=1e-3,
               keep_weights=True):
    raw = synthetic_series(T, N, 303)
    train_data, valid_data = raw[:ntrain], raw[ntrain:]
    ref_model_cls = hd.Model
    log = dict(order=[], losses=[], metrics=[], init=None)

    def make_model(*a, **k):
        m = ref_model_cls(*a, dropout_rate=0.0, **k)            # handler.py:105 never forwards dropout; pin it to 0
        log["init"] = {kk: v.detach().clone().numpy() for kk, v in m.state_dict().items()}
        return m

    class LoggedDataset(fd.ForecastDataset):
        def __getitem__(self, index):
            log["order"].append((id(self), int(index)))
            return super().__getitem__(index)

    class LoggedMSE(torch.nn.MSELoss):
        def forward(self, a, b):
            v = super().forward(a, b)
            log["losses"].append(float(v.detach()))
            return v

    real_validate = hd.validate

    def logged_validate(*a, **k):
        r = real_validate(*a, **k)
        log["metrics"].append(r)
        return r

    saved = (hd.Model, hd.ForecastDataset, hd.validate, hd.nn)
    hd.Model, hd.ForecastDataset, hd.validate = make_model, LoggedDataset, logged_validate
    hd.nn = types.SimpleNamespace(MSELoss=LoggedMSE)
    args = types.SimpleNamespace(window_size=W, horizon=H, multi_layer=multi, device="cpu", norm_method="z_score",
                                 optimizer="RMSProp", lr=lr, decay_rate=0.5, exponential_decay_step=2,
                                 batch_size=bs, epoch=epochs, validate_freq=1, early_stop=False)
    torch.manual_seed(0)                                        # main.py:52
    try:
        with tempfile.TemporaryDirectory() as d:
            metrics, stat = hd.train(train_data, valid_data, args, d)
            with open(os.path.join(d, "_stemgnn.pt"), "rb") as f:   # best model (handler.py:184-185); shim item 4
                final = torch.load(f, weights_only=False)
    finally:
        hd.Model, hd.ForecastDataset, hd.validate, hd.nn = saved
    ids = []
    for i, _ in log["order"]:
        if i not in ids:
            ids.append(i)
    train_order = np.asarray([ix for i, ix in log["order"] if i == ids[0]], np.int64)
    out = dict(cfg=np.array([T, N, W, H, multi, bs, epochs, ntrain], np.int64), lr=np.float64(lr),
               train_order=train_order, losses=np.asarray(log["losses"], np.float64),
               stat_mean=np.asarray(stat["mean"]), stat_std=np.asarray(stat["std"]))
    for e, m in enumerate(log["metrics"]):
        for k, v in m.items():
            out[f"val{e}_{k}"] = np.asarray(v, np.float64)
    if keep_weights:                       # small case: the series and both state dicts travel with the fixture
        out["raw"] = raw
        for k, v in log["init"].items():
            out["init." + k] = v
        for k, v in final.state_dict().items():
            out["final." + k] = v.detach().numpy()
    else:                                  # real shape: series and initial weights are regenerated from the seeds
        out["raw_seed"] = np.int64(303)
        out["init_sum"] = np.float64(sum(float(np.abs(v).sum()) for v in log["init"].values()))
    save(name, **out)
    print("  losses", np.round(log["losses"][:3], 5), "...", np.round(log["losses"][-2:], 5),
          "val mae", [float(m["mae"]) for m in log["metrics"]])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    fd, mu, hd = load_reference_packages()
    case_norm(fd, "z_score")
    case_norm(fd, "min_max")
    case_metrics(fd, mu)
    case_rolling(fd, hd)
    case_train(fd, hd)
    # the headline shape (PEMS07: N=228, W=12, H=3, multi=5, batch 32): 2 epochs of the reference's handler.train
    case_train(fd, hd, name="train_pems07", T=500, N=228, W=12, H=3, multi=5, bs=32, epochs=2, ntrain=350, lr=1e-4,
               keep_weights=False)
