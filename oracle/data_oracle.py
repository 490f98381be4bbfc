"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference's data path either side of the hot path.

Covers SURVEY.md section 8f rows 3 and 4:
  * normalisation / de-normalisation           data_loader/forecast_dataloader.py:7-38
  * window indexing of ``ForecastDataset``      data_loader/forecast_dataloader.py:41-73
  * rolling inference                           models/handler.py:41-65
  * MAPE / MAE / RMSE / ``evaluate``            utils/math_utils.py:24-74
Pinned by ``tests/golden/data_*.npz`` (outputs of the real reference, ``tests/golden/make_golden_data.py``).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np


def fill_na(data):
    """forward-fill then backward-fill along time (forecast_dataloader.py:49)."""
    data = np.array(data, dtype=np.float64, copy=True)
    T = data.shape[0]
    for t in range(1, T):
        row = data[t]
        bad = np.isnan(row)
        row[bad] = data[t - 1][bad]
    for t in range(T - 2, -1, -1):
        row = data[t]
        bad = np.isnan(row)
        row[bad] = data[t + 1][bad]
    return data


def normalized(data, normalize_method, norm_statistic=None):
    """forecast_dataloader.py:7-22 (float64 arithmetic; eps 1e-5; clip to [0,1]; std==0 -> 1)."""
    data = np.asarray(data, dtype=np.float64)
    if normalize_method == "min_max":
        if not norm_statistic:
            norm_statistic = dict(max=np.max(data, axis=0), min=np.min(data, axis=0))
        lo = np.asarray(norm_statistic["min"], dtype=np.float64)
        scale = np.asarray(norm_statistic["max"], dtype=np.float64) - lo + 1e-5
        data = np.clip((data - lo) / scale, 0.0, 1.0)
    elif normalize_method == "z_score":
        if not norm_statistic:
            norm_statistic = dict(mean=np.mean(data, axis=0), std=np.std(data, axis=0))
        mean = np.asarray(norm_statistic["mean"], dtype=np.float64)
        std = np.asarray([1 if s == 0 else s for s in norm_statistic["std"]], dtype=np.float64)
        data = (data - mean) / std
        norm_statistic = dict(norm_statistic, std=list(std))
    return data, norm_statistic


def de_normalized(data, normalize_method, norm_statistic):
    """forecast_dataloader.py:25-38 (note the 1e-8 eps here against 1e-5 forward)."""
    data = np.asarray(data)
    if normalize_method == "min_max":
        lo = np.asarray(norm_statistic["min"], dtype=np.float64)
        scale = np.asarray(norm_statistic["max"], dtype=np.float64) - lo + 1e-8
        data = data * scale + lo
    elif normalize_method == "z_score":
        mean = np.asarray(norm_statistic["mean"], dtype=np.float64)
        std = np.asarray([1 if s == 0 else s for s in norm_statistic["std"]], dtype=np.float64)
        data = data * std + mean
    return data


def x_end_idx(T, window_size, horizon, interval=1):
    """forecast_dataloader.py:68-73."""
    xs = list(range(window_size, T - horizon + 1))
    return [xs[j * interval] for j in range(len(xs) // interval)]


def window(data, hi, window_size, horizon):
    """forecast_dataloader.py:56-63 (`.type(torch.float)` = float32 cast)."""
    return (np.asarray(data[hi - window_size:hi], dtype=np.float32),
            np.asarray(data[hi:hi + horizon], dtype=np.float32))


def batches(data, order, batch_size, window_size, horizon, drop_last=False):
    """default collate of DataLoader over `order` (a list of x_end_idx values)."""
    for s in range(0, len(order), batch_size):
        chunk = order[s:s + batch_size]
        if drop_last and len(chunk) < batch_size:
            return
        xs, ys = zip(*(window(data, hi, window_size, horizon) for hi in chunk))
        yield np.stack(xs), np.stack(ys)


def rolling_inference(model_fn, inputs, window_size, horizon):
    """models/handler.py:47-63 for ONE batch.  model_fn(inputs[B,W,N]) -> forecast[B,L,N] (float32)."""
    inputs = np.array(inputs, dtype=np.float32, copy=True)
    B, _, N = inputs.shape
    forecast_steps = np.zeros([B, horizon, N], dtype=np.float64)
    step = 0
    while step < horizon:
        out = np.asarray(model_fn(inputs), dtype=np.float32)
        L = out.shape[1]
        if L == 0:
            raise Exception("Get blank inference result")
        inputs[:, :window_size - L, :] = inputs[:, L:window_size, :].copy()
        inputs[:, window_size - L:, :] = out
        take = min(horizon - step, L)
        forecast_steps[:, step:step + take, :] = out[:, :take, :]
        step += take
    return forecast_steps


def MAPE(v, v_, axis=None):
    """utils/math_utils.py:24-34: `+1e-5` is added to the ratio, clipped at 5 (NaN passes through)."""
    mape = (np.abs(v_ - v) / np.abs(v) + 1e-5).astype(np.float64)
    mape = np.where(mape > 5, 5, mape)
    return np.mean(mape, axis)


def RMSE(v, v_, axis=None):
    """utils/math_utils.py:37-45."""
    return np.sqrt(np.mean((v_ - v) ** 2, axis)).astype(np.float64)


def MAE(v, v_, axis=None):
    """utils/math_utils.py:48-56."""
    return np.mean(np.abs(v_ - v), axis).astype(np.float64)


def evaluate(y, y_hat, by_step=False, by_node=False):
    """utils/math_utils.py:59-74.  y, y_hat: [count, time_step, node]."""
    if by_step and by_node:
        ax = 0
    elif by_step:
        ax = (0, 2)
    elif by_node:
        ax = (0, 1)
    else:
        ax = None
    return MAPE(y, y_hat, ax), MAE(y, y_hat, ax), RMSE(y, y_hat, ax)


def train_restated(train_data, valid_data, init_sd, orders, window_size, horizon, multi_layer, batch_size, epochs,
                   lr, decay_rate, exponential_decay_step, dropout_rate=0.0, dtype=None):
    """models/handler.py:103-192 restated on the oracle model (z_score, RMSProp, validate every epoch).

    `orders`: per-epoch lists of dataset indices (the shuffle the reference drew).  Returns per-step losses,
    per-epoch validation dicts (mae, mape, rmse, mae_node ...) and the final parameters.
    """
    import torch
    from collections import OrderedDict
    from .stemgnn_oracle import model_forward

    dtype = dtype or torch.float32
    stat = {"mean": np.mean(train_data, axis=0).tolist(), "std": np.std(train_data, axis=0).tolist()}    # :112-115
    tr, _ = normalized(fill_na(train_data), "z_score", stat)
    va, _ = normalized(fill_na(valid_data), "z_score", stat)
    tr_idx = x_end_idx(len(tr), window_size, horizon)
    va_idx = x_end_idx(len(va), window_size, horizon)
    sd = OrderedDict((k, torch.as_tensor(v).to(dtype).clone().requires_grad_(True)) for k, v in init_sd.items())
    opt = torch.optim.RMSprop(list(sd.values()), lr=lr, eps=1e-8)                                       # :127
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=decay_rate)                                # :130
    losses, metrics = [], []
    for epoch in range(epochs):
        order = [tr_idx[i] for i in orders[epoch]]
        for xb, yb in batches(tr, order, batch_size, window_size, horizon):                              # :157-166
            opt.zero_grad(set_to_none=True)
            mask = None
            if dropout_rate > 0.0:
                N = xb.shape[2]
                mask = (torch.rand(xb.shape[0], N, N) >= dropout_rate).to(dtype)
            f, _ = model_forward(torch.from_numpy(xb).to(dtype), sd, drop_mask=mask, drop_p=dropout_rate)
            loss = torch.nn.functional.mse_loss(f, torch.from_numpy(yb).to(dtype))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        if (epoch + 1) % exponential_decay_step == 0:                                                     # :170-171
            sched.step()

        def fwd(inp):
            with torch.no_grad():
                return model_forward(torch.from_numpy(inp).to(dtype), sd)[0].float().numpy()

        fs, ts = [], []
        for xb, yb in batches(va, va_idx, batch_size, window_size, horizon):                              # :41-65
            fs.append(rolling_inference(fwd, xb, window_size, horizon))
            ts.append(yb)
        f_norm, t_norm = np.concatenate(fs), np.concatenate(ts)
        f_raw, t_raw = de_normalized(f_norm, "z_score", stat), de_normalized(t_norm, "z_score", stat)     # :73-75
        s, sn = evaluate(t_raw, f_raw), evaluate(t_raw, f_raw, by_node=True)
        metrics.append(dict(mape=s[0], mae=s[1], rmse=s[2], mape_node=sn[0], mae_node=sn[1], rmse_node=sn[2]))
    return losses, metrics, OrderedDict((k, v.detach()) for k, v in sd.items())
