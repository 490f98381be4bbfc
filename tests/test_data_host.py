"""Host-side logic of the device data path (no GPU needed): NaN filling, statistics, epoch order / RNG parity with
torch's DataLoader and with the batch order the REAL reference drew (tests/golden/data/train_e2e.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import data_oracle as do
from stemgnn_amd import _lib
from stemgnn_amd import forecast_dataloader as fdl
from tests.util import GOLDEN_DIR


def G(name):
    return np.load(os.path.join(GOLDEN_DIR, "data", name + ".npz"))


def test_fill_na_matches_oracle():
    rng = np.random.default_rng(3)
    for T, N, frac in [(1, 3, 0.5), (17, 5, 0.3), (40, 6, 0.7), (25, 4, 0.0)]:
        a = rng.normal(size=(T, N))
        a[rng.random((T, N)) < frac] = np.nan
        if T > 1:
            a[:, 0] = np.nan                                            # a column with no valid value stays NaN
        np.testing.assert_array_equal(fdl._fill_na(a), do.fill_na(a))
    z = G("norm_z_score")
    np.testing.assert_array_equal(fdl._fill_na(z["raw"][:28]), z["filled_train"])


@pytest.mark.parametrize("method", ["z_score", "min_max"])
def test_statistics_match_oracle(method):
    z = G("norm_" + method)
    raw = do.fill_na(z["raw"])
    keys = ("mean", "std") if method == "z_score" else ("min", "max")
    for stat in ({k: z["stat_" + k].tolist() for k in keys}, None):
        sub, div, clip, _ = fdl._stat_arrays(raw, method, stat)
        want = z["data"] if stat else z["data_ownstat"]
        got = (raw - sub) / div
        if clip:
            got = np.clip(got, 0.0, 1.0)
        np.testing.assert_array_equal(got, want)


def test_epoch_order_consumes_rng_like_dataloader():
    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 37

        def __getitem__(self, i):
            return torch.tensor(i)

    torch.manual_seed(5)
    tl = torch.utils.data.DataLoader(DS(), batch_size=8, shuffle=True, num_workers=0)
    vl = torch.utils.data.DataLoader(DS(), batch_size=8, shuffle=False, num_workers=0)
    ref = []
    for _ in range(3):
        ref.append(torch.cat(list(tl)).tolist())
        ref.append(torch.cat(list(vl)).tolist())
    tail_ref = torch.rand(4)
    torch.manual_seed(5)
    got = []
    for _ in range(3):
        got.append(fdl.epoch_order(37, True))
        got.append(fdl.epoch_order(37, False))
    assert got == ref
    assert torch.equal(torch.rand(4), tail_ref)                         # same RNG position afterwards


def test_seed_parity_with_reference_run():
    """torch.manual_seed(0) -> Model(...) -> loaders: the drop-in draws the same initial weights and the same three
    epochs of batch order as the reference's handler.train did."""
    from stemgnn_amd import Model
    z = G("train_e2e")
    T, N, W, H, multi, bs, epochs, ntrain = (int(v) for v in z["cfg"])
    torch.manual_seed(0)
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0)
    for k, v in model.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), z["init." + k], err_msg=k)
    n_tr, n_va = len(do.x_end_idx(ntrain, W, H)), len(do.x_end_idx(T - ntrain, W, H))
    order = []
    for _ in range(epochs):
        order += fdl.epoch_order(n_tr, True)
        fdl.epoch_order(n_va, False)
    assert order == z["train_order"].tolist()


def test_data_path_refuses_cpu():
    with pytest.raises(_lib.StemGNNHipError):
        fdl.ForecastDataset(np.zeros((30, 3)), 5, 2, device="cpu")
    with pytest.raises(_lib.StemGNNHipError):
        fdl.normalized(np.zeros((30, 3)), "z_score", device="cpu")


def test_data_entry_points_reject_bad_arguments():
    lib = _lib.load()
    p = 4096                                                            # fake non-null pointer: checks come first
    assert lib.stemgnn_roll_window(p, p + 64, p + 128, p + 192, 2, 4, 6, 3, 0, 5, None) == _lib.SG_EINVAL   # L > W
    assert lib.stemgnn_roll_window(p, p + 64, p, p + 192, 2, 4, 2, 3, 0, 5, None) == _lib.SG_EINVAL         # in place
    assert lib.stemgnn_window_gather(p, p, p, p, 2, 12, 3, 5, 10, None, None) == _lib.SG_EINVAL             # T < W+H
    assert lib.stemgnn_eval_metrics(p, p, p, None, 5, 3, 4, p, p, None) == _lib.SG_EINVAL                   # mul w/o add
    assert lib.stemgnn_mse_fwd(None, p, 8, p, p, None, None) == _lib.SG_EINVAL
    assert lib.stemgnn_normalize_series(p, p, p, 0, p, 0, 4, None) == _lib.SG_EINVAL
    assert lib.stemgnn_eval_out_doubles(3, 5) == 3 + 15 + 9 + 45
    assert lib.stemgnn_eval_scratch_doubles(130, 3, 5) == 3 * 15 * (3 + 1) + 15


def test_checkpoint_names_follow_the_reference():
    """models/handler.py:21-22 names a snapshot ``str(epoch) if epoch else ''`` + '_stemgnn.pt': epoch 0 and the best model
    share '_stemgnn.pt' (ADVICE round 2)."""
    from stemgnn_amd.trainer import checkpoint_path

    assert checkpoint_path("out").name == "_stemgnn.pt"
    assert checkpoint_path("out", 0).name == "_stemgnn.pt"
    assert checkpoint_path("out", 7).name == "7_stemgnn.pt"
