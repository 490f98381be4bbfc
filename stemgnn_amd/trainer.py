"""Epoch driver over the device-side components -- an original, minimal loop, NOT a copy of the
reference's control plane.

The reference's driver (``models/handler.py``) is out of scope; what the hot path needs either side of it
are four device components, and this module only strings them together:

    ForecastDataset / WindowLoader   resident [T,N] series, index batches        forecast_dataloader.py
    engine.TrainStep                 gather -> fwd -> MSE -> bwd -> optimizer     one hipGraph per step
    ops.roll_window                  rolling multi-step inference                 csrc/data.hip
    math_utils.Scores                de-normalise + MAPE / MAE / RMSE             csrc/data.hip

Two ways to use it:
  * ``DeviceTrainer(...).fit(train_series, valid_series, epochs)`` / ``.evaluate(series)`` -- the native API;
  * ``train(train_data, valid_data, args, result_dir)`` / ``test(test_data, args, train_dir, test_dir)`` -- adapters with
    the call shape ``main.py`` uses (main.py:5,54,60), so the reference's entry script can drive the device loop by
    importing these two names instead of ``models.handler``'s.  Checkpoints use the reference's file names
    (``<epoch>_stemgnn.pt``, best model ``_stemgnn.pt``, whole-module pickles) so they interchange.
The unmodified reference driver also works on top of ``stemgnn_amd.Model`` alone.
"""
import json
import pathlib
import time

import numpy as np
import torch

from . import ops
from .base_model import Model
from .engine import TrainStep
from .forecast_dataloader import ForecastDataset, WindowLoader, denorm_coefficients
from .math_utils import Scores
from .optim import FusedAdam, FusedRMSprop

BEST = "_stemgnn.pt"


def checkpoint_path(directory, tag=None):
    """The reference's file names (models/handler.py:21-22: ``str(epoch) if epoch else ''``): the snapshot of epoch 0
    shares the best-model slot ``_stemgnn.pt`` there, so a reference loader finds a model after a one-epoch run without
    validation; kept identical so the files interchange."""
    return pathlib.Path(directory) / (f"{tag}{BEST}" if tag else BEST)


def save_checkpoint(model, directory, tag=None):
    path = checkpoint_path(directory, tag)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(model, path)
    return path


def load_checkpoint(directory, tag=None):
    path = checkpoint_path(directory, tag)
    return torch.load(path, weights_only=False) if path.is_file() else None


def rolling_forecast(model, loader, horizon):
    """Multi-step forecast of every window the loader yields: the model's first outputs are fed back as inputs until
    `horizon` steps exist (what the reference's validation does, models/handler.py:41-65), all on the device.
    Returns (forecast [count, horizon, N], target [count, horizon, N]) float32 device tensors."""
    was_training = model.training
    model.eval()
    forecasts, targets = [], []
    with torch.no_grad():
        for window, target in loader:
            steps = torch.zeros(window.shape[0], horizon, window.shape[2], device=window.device)
            done = 0
            while done < horizon:
                out, _ = model(window)
                if out.shape[1] == 0:
                    raise Exception("Get blank inference result")
                window = ops.roll_window(window, out, steps, done, horizon)
                done += min(horizon - done, out.shape[1])
            forecasts.append(steps)
            targets.append(target)
    model.train(was_training)
    return torch.cat(forecasts), torch.cat(targets)


def score_forecast(forecast, target, norm_method=None, statistic=None, dump_dir=None):
    """Metrics of a rolling forecast in raw units (and normalised units under ``*_norm``).  With `dump_dir`, the first
    forecast step of every window is written as CSV (target / predict / absolute error / absolute percentage error)."""
    mul = add = None
    if norm_method and statistic:
        mul, add = denorm_coefficients(norm_method, statistic, forecast.device)
    raw = Scores(target, forecast, mul, add)
    (mape, mae, rmse), (mape_n, mae_n, rmse_n) = raw.get(), raw.get(by_node=True)
    out = dict(mae=mae, mape=mape, rmse=rmse, mae_node=mae_n, mape_node=mape_n, rmse_node=rmse_n)
    normed = Scores(target, forecast).get() if mul is not None else (mape, mae, rmse)
    out.update(mape_norm=normed[0], mae_norm=normed[1], rmse_norm=normed[2])
    if dump_dir is not None:
        d = pathlib.Path(dump_dir)
        d.mkdir(parents=True, exist_ok=True)
        pred, true = forecast[:, 0, :].double(), target[:, 0, :].double()
        if mul is not None:
            pred, true = pred * mul + add, true * mul + add
        pred, true = pred.cpu().numpy(), true.cpu().numpy()
        err = np.abs(pred - true)
        with np.errstate(divide="ignore", invalid="ignore"):
            ape = err / np.abs(true)
        for name, arr in (("target", true), ("predict", pred), ("predict_abs_error", err), ("predict_ape", ape)):
            np.savetxt(d / f"{name}.csv", arr, delimiter=",")
    return out


def column_statistics(series, norm_method):
    series = np.asarray(series)
    if norm_method == "z_score":
        return {"mean": series.mean(axis=0).tolist(), "std": series.std(axis=0).tolist()}
    if norm_method == "min_max":
        return {"min": series.min(axis=0).tolist(), "max": series.max(axis=0).tolist()}
    return None


class DeviceTrainer:
    """Owns model + optimizer + LR schedule; `fit` runs epochs of hipGraph train steps with validation in between."""

    def __init__(self, units, window, horizon, multi, *, batch_size=32, lr=1e-4, optimizer="RMSProp", decay_rate=0.5,
                 decay_every=5, norm_method="z_score", device="cuda", model_factory=None, hipgraph=True,
                 dropout_seed=None):
        self.units, self.window, self.horizon, self.multi = units, window, horizon, multi
        self.batch_size, self.norm_method, self.device, self.hipgraph = batch_size, norm_method, device, hipgraph
        self.decay_every = decay_every
        self.model = (model_factory or Model)(units, 2, window, multi, horizon=horizon)
        self.model.to(device)
        if dropout_seed is not None and hasattr(self.model, "set_dropout_seed"):
            # an explicit Philox key for the attention dropout: by default the key follows the device generator's seed AND
            # the model's construction index in this process (two models never share a mask stream), so a run is only
            # reproducible if models are built in the same order; naming the key removes that dependence
            self.model.set_dropout_seed(int(dropout_seed), 0, torch.device(device))
        if optimizer == "RMSProp":
            self.optimizer = FusedRMSprop(self.model.parameters(), lr=lr, eps=1e-8)
        else:                                   # the driver's other branch (models/handler.py:128-129), fused as well
            self.optimizer = FusedAdam(self.model.parameters(), lr=lr, betas=(0.9, 0.999))
        self.schedule = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=decay_rate)
        self.statistic = None
        self.stepper = None

    def _dataset(self, series):
        return ForecastDataset(series, window_size=self.window, horizon=self.horizon, normalize_method=self.norm_method,
                               norm_statistic=self.statistic, device=self.device)

    def validate(self, loader, dump_dir=None):
        forecast, target = rolling_forecast(self.model, loader, self.horizon)
        return score_forecast(forecast, target, self.norm_method, self.statistic, dump_dir)

    def fit(self, train_series, valid_series, epochs, *, validate_every=1, patience=None, out_dir=None, on_step=None,
            on_validate=None, log=print):
        if len(train_series) == 0:
            raise Exception("Cannot organize enough training data")
        if len(valid_series) == 0:
            raise Exception("Cannot organize enough validation data")
        self.statistic = column_statistics(train_series, self.norm_method)
        if out_dir is not None and self.statistic is not None:
            pathlib.Path(out_dir).mkdir(parents=True, exist_ok=True)
            (pathlib.Path(out_dir) / "norm_stat.json").write_text(json.dumps(self.statistic))
        train_set, valid_set = self._dataset(train_series), self._dataset(valid_series)
        batches = WindowLoader(train_set, batch_size=self.batch_size, drop_last=False, shuffle=True)
        valid_loader = WindowLoader(valid_set, batch_size=self.batch_size, shuffle=False)
        log(f"trainable parameters: {sum(p.numel() for p in self.model.parameters() if p.requires_grad)}")
        self.stepper = TrainStep(self.model, self.optimizer, self.batch_size, self.window, self.horizon, self.units,
                                 series=train_set.data, graph=self.hipgraph, order_capacity=len(train_set))
        best, stale, metrics = float("inf"), 0, {}
        for epoch in range(epochs):
            t0 = time.time()
            self.model.train()
            n_steps = 0
            # the epoch's shuffled order goes to the device-side iterator once; full batches are then one graph replay
            # each (TrainStep.run_next), the ragged last batch (drop_last=False, handler.py:136) runs eagerly
            epoch_batches = list(batches.index_batches())
            full = [idx for idx in epoch_batches if idx.numel() == self.batch_size]
            if full:
                self.stepper.load_order(train_set.hi_all.index_select(0, torch.cat(full)))
            for i, idx in enumerate(epoch_batches):
                if idx.numel() == self.batch_size:
                    self.stepper.run_next()
                else:
                    self.stepper.run_indices(train_set.hi_all.index_select(0, idx))
                n_steps += 1
                if on_step is not None:
                    on_step(epoch, i, self.stepper)
            mean_loss = self.stepper.epoch_loss_sum() / max(n_steps, 1)       # one host sync per epoch
            ops.check_gather_status(train_set.device)                       # both raise if a device-side check tripped
            ops.check_gru_status(train_set.device)
            log(f"epoch {epoch}: {time.time() - t0:.2f}s  mean train loss {mean_loss:.4f}  [{self.stepper.mode}]")
            if out_dir is not None:
                save_checkpoint(self.model, out_dir, epoch)
            if (epoch + 1) % self.decay_every == 0:
                self.schedule.step()
            if (epoch + 1) % validate_every == 0:
                metrics = self.validate(valid_loader, out_dir)
                ops.check_gru_status(train_set.device)
                log(f"  validation: MAPE {metrics['mape']:.6%}  MAE {metrics['mae']:.6f}  RMSE {metrics['rmse']:.6f}"
                    f"  (normalised MAE {metrics['mae_norm']:.6f})")
                if on_validate is not None:
                    on_validate(epoch, metrics)
                if metrics["mae"] < best:
                    best, stale = metrics["mae"], 0
                    if out_dir is not None:
                        save_checkpoint(self.model, out_dir)
                else:
                    stale += 1
            if patience is not None and stale >= patience:
                break
        return metrics, self.statistic

    def evaluate(self, series, dump_dir=None):
        loader = WindowLoader(self._dataset(series), batch_size=self.batch_size, drop_last=False, shuffle=False)
        return self.validate(loader, dump_dir)


# ---- adapters with main.py's call shape (main.py:54, :60) -----------------------------------------------------------
def train(train_data, valid_data, args, result_file, model_factory=None, on_step=None, on_validate=None):
    trainer = DeviceTrainer(train_data.shape[1], args.window_size, args.horizon, args.multi_layer,
                            batch_size=args.batch_size, lr=args.lr, optimizer=args.optimizer, decay_rate=args.decay_rate,
                            decay_every=args.exponential_decay_step, norm_method=args.norm_method, device=args.device,
                            model_factory=model_factory, hipgraph=getattr(args, "hipgraph", True))
    patience = getattr(args, "early_stop_step", 10) if getattr(args, "early_stop", False) else None
    return trainer.fit(train_data, valid_data, args.epoch, validate_every=args.validate_freq, patience=patience,
                       out_dir=result_file, on_step=on_step, on_validate=on_validate)


def test(test_data, args, result_train_file, result_test_file):
    model = load_checkpoint(result_train_file)
    if model is None:
        raise FileNotFoundError(f"no best-model checkpoint under {result_train_file}")
    statistic = json.loads((pathlib.Path(result_train_file) / "norm_stat.json").read_text())
    dataset = ForecastDataset(test_data, window_size=args.window_size, horizon=args.horizon,
                              normalize_method=args.norm_method, norm_statistic=statistic, device=args.device)
    loader = WindowLoader(dataset, batch_size=args.batch_size, drop_last=False, shuffle=False)
    forecast, target = rolling_forecast(model, loader, args.horizon)
    metrics = score_forecast(forecast, target, args.norm_method, statistic, result_test_file)
    print(f"test: MAPE {metrics['mape']:.4f}  MAE {metrics['mae']:.4f}  RMSE {metrics['rmse']:.4f}")
    return metrics
