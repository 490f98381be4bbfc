"""The ONE-IMPORT integration of INTEGRATION.md section 1, run for real on the GPU: the body of the reference driver's
training loop (models/handler.py:152-171) and of its validation loop (``inference``, :41-65; ``validate``, :68-100) re-typed
here with STOCK torch pieces -- ``torch.utils.data.DataLoader(shuffle=True, num_workers=0)``, ``torch.optim.RMSprop(lr,
eps=1e-8)``, ``torch.optim.lr_scheduler.ExponentialLR``, ``nn.MSELoss``, ``model.zero_grad()``, ``float(loss)``, the
in-place ``inputs[...] =`` window roll -- around ``stemgnn_amd.Model`` as the only non-reference object.  None of the
package's own loop machinery (trainer / TrainStep / FusedRMSprop / WindowLoader / hipGraph) is involved.

Pinned to ``tests/golden/data/train_e2e.npz``: the reference's own ``handler.train`` run (3 epochs, dropout 0) -- same
torch seed -> same initial weights and the same DataLoader shuffle; per-step losses within 1e-3, validation
MAPE / MAE / RMSE within 2e-3.  The host-side dataset below is test infrastructure on top of ``oracle.data_oracle``
(the reference's ForecastDataset cannot travel to the GPU box).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.utils.data as torch_data

import os

from oracle import data_oracle as do
from tests.util import GOLDEN_DIR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def G(name):
    return np.load(os.path.join(GOLDEN_DIR, "data", name + ".npz"))


class HostForecastDataset(torch_data.Dataset):
    """CPU dataset with the reference's item semantics (data_loader/forecast_dataloader.py:41-73): float64 normalised
    series on the host, one (x [W,N], y [H,N]) float32 CPU pair per index."""

    def __init__(self, df, window_size, horizon, normalize_method=None, norm_statistic=None, interval=1):
        self.window_size, self.horizon = window_size, horizon
        data = do.fill_na(np.asarray(df, dtype=np.float64))
        self.data, _ = do.normalized(data, normalize_method, norm_statistic)
        self.x_end_idx = do.x_end_idx(len(data), window_size, horizon, interval)

    def __len__(self):
        return len(self.x_end_idx)

    def __getitem__(self, index):
        x, y = do.window(self.data, self.x_end_idx[index], self.window_size, self.horizon)
        return torch.from_numpy(x), torch.from_numpy(y)


def reference_inference(model, dataloader, device, node_cnt, window_size, horizon):
    """models/handler.py:41-65, line for line in behaviour (np.float -> float for numpy >= 1.24)."""
    forecast_set, target_set = [], []
    model.eval()
    with torch.no_grad():
        for inputs, target in dataloader:
            inputs = inputs.to(device)
            target = target.to(device)
            step = 0
            forecast_steps = np.zeros([inputs.size()[0], horizon, node_cnt], dtype=float)
            while step < horizon:
                forecast_result, _ = model(inputs)
                n_out = forecast_result.size()[1]
                if n_out == 0:
                    raise Exception("Get blank inference result")
                inputs[:, :window_size - n_out, :] = inputs[:, n_out:window_size, :].clone()      # in-place roll (:56-58)
                inputs[:, window_size - n_out:, :] = forecast_result.clone()
                take = min(horizon - step, n_out)
                forecast_steps[:, step:take + step, :] = forecast_result[:, :take, :].detach().cpu().numpy()
                step += take
            forecast_set.append(forecast_steps)
            target_set.append(target.detach().cpu().numpy())
    return np.concatenate(forecast_set, axis=0), np.concatenate(target_set, axis=0)


def reference_validate(model, loader, device, method, stat, node_cnt, window_size, horizon):
    """models/handler.py:68-100 without the CSV dumps."""
    forecast_norm, target_norm = reference_inference(model, loader, device, node_cnt, window_size, horizon)
    forecast = do.de_normalized(forecast_norm, method, stat)
    target = do.de_normalized(target_norm, method, stat)
    score = do.evaluate(target, forecast)
    by_node = do.evaluate(target, forecast, by_node=True)
    return dict(mape=score[0], mae=score[1], rmse=score[2], mape_node=by_node[0], mae_node=by_node[1],
                rmse_node=by_node[2])


def test_reference_driver_loop_on_the_drop_in():
    from stemgnn_amd import Model
    from stemgnn_amd import ops
    z = G("train_e2e")
    T, N, W, H, multi, bs, epochs, ntrain = (int(v) for v in z["cfg"])
    raw = z["raw"]
    train_data, valid_data = raw[:ntrain], raw[ntrain:]
    torch.manual_seed(0)                                                      # main.py:52
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0)                # handler.py:105 (fixture: dropout 0)
    model.to(DEV)                                                             # :106
    for k, v in model.state_dict().items():                                   # same seed -> the reference's initial weights
        np.testing.assert_array_equal(v.cpu().numpy(), z["init." + k], err_msg=k)
    stat = {"mean": np.mean(train_data, axis=0).tolist(), "std": np.std(train_data, axis=0).tolist()}     # :112-114
    my_optim = torch.optim.RMSprop(params=model.parameters(), lr=float(z["lr"]), eps=1e-08)               # :127
    sched = torch.optim.lr_scheduler.ExponentialLR(optimizer=my_optim, gamma=0.5)                         # :130
    train_set = HostForecastDataset(train_data, W, H, "z_score", dict(stat))
    valid_set = HostForecastDataset(valid_data, W, H, "z_score", dict(stat))
    train_loader = torch_data.DataLoader(train_set, batch_size=bs, drop_last=False, shuffle=True, num_workers=0)   # :136-137
    valid_loader = torch_data.DataLoader(valid_set, batch_size=bs, shuffle=False, num_workers=0)                   # :138
    forecast_loss = nn.MSELoss(reduction="mean").to(DEV)                      # :140
    assert not model.hot_state.direct and not model.hot_state.overlap         # stock optimizer: plain autograd gradients

    losses, vals = [], []
    for epoch in range(epochs):
        model.train()                                                         # :154 (also the GRU cluster health check)
        for inputs, target in train_loader:
            inputs = inputs.to(DEV)                                           # :158-159
            target = target.to(DEV)
            model.zero_grad()                                                 # :160 (set_to_none: grads dropped each step)
            forecast, _ = model(inputs)                                       # :161
            loss = forecast_loss(forecast, target)                            # :162
            loss.backward()                                                   # :164
            my_optim.step()                                                   # :165
            losses.append(float(loss))                                        # :166
        if (epoch + 1) % 2 == 0:                                              # :171-172 (fixture: decay step 2)
            sched.step()
        vals.append(reference_validate(model, valid_loader, DEV, "z_score", stat, N, W, H))   # :173-178

    np.testing.assert_allclose(np.asarray(losses), z["losses"], rtol=1e-3)
    for e in range(epochs):
        for k in ("mae", "mape", "rmse", "mae_node", "rmse_node"):
            np.testing.assert_allclose(vals[e][k], z[f"val{e}_{k}"], rtol=2e-3, err_msg=f"epoch {e} {k}")
    # block 1's short-cut never receives a gradient (reference :73-74): stock zero_grad/backward leaves it None
    assert model.stock_block[1].backcast_short_cut.weight.grad is None
    assert all(p.grad is not None for n, p in model.named_parameters() if "stock_block.1.backcast_short_cut" not in n)
    ops.check_gru_status(torch.device(DEV))


def test_model_train_toggle_reports_a_lost_gru_partner():
    """Model.train()/eval() is where the reference driver would learn about a GRU cluster time-out (one host sync per
    toggle): a status word set by the kernels raises there, once."""
    from stemgnn_amd import Model, ops
    from stemgnn_amd._lib import StemGNNHipError
    N, W = 16, 6
    model = Model(N, 2, W, 2, horizon=3).to(DEV)
    model.eval()
    with torch.no_grad():
        model(torch.randn(4, W, N, device=DEV))
    model.train()                                                             # healthy: no raise
    ops.gru_status(torch.device(DEV)).fill_(1)                                # what a timed-out exchange leaves behind
    with pytest.raises(StemGNNHipError):
        model.eval()
    model.train()                                                             # reported once, then clear again
    assert not model.training or model.training is True


@pytest.mark.parametrize("stack_cnt", [1, 2, 3])
def test_stack_count_behaves_like_the_reference(stack_cnt):
    """The reference constructor builds any number of blocks (:93-95); its forward only works for 2 (:171-174): 3+ fails
    on block 1's None backcast, 1 on result[1]."""
    from stemgnn_amd import Model
    N, W = 12, 6
    model = Model(N, stack_cnt, W, 2, horizon=3, device=DEV)
    assert len(model.stock_block) == stack_cnt
    assert hasattr(model.stock_block[0], "backcast") and all(not hasattr(b, "backcast") for b in model.stock_block[1:])
    x = torch.randn(2, W, N, device=DEV)
    if stack_cnt == 2:
        out, att = model(x)
        assert out.shape == (2, 3, N) and att.shape == (N, N)
    else:
        with pytest.raises(AttributeError if stack_cnt > 2 else IndexError):
            model(x)
