"""Device-resident mirror of the reference's ``data_loader/forecast_dataloader.py``.

Same names and argument meaning: ``normalized``, ``de_normalized``, ``ForecastDataset``.  The normalised series lives in
HBM as ONE [T, N] fp32 matrix; a sample / batch is gathered from it by index with ``stemgnn_window_gather`` -- no
per-sample ``from_numpy``, collate or H2D copy inside the step (reference: forecast_dataloader.py:56-63 +
models/handler.py:136-138,158-159).  ``WindowLoader`` stands in for the ``torch_data.DataLoader`` the driver wraps the
dataset in, drawing its shuffle from the torch RNG exactly as DataLoader + RandomSampler do (same batches for the same
``torch.manual_seed``).

Load-time host work (CSV -> numpy, NaN filling, column statistics) stays on the host as in the reference; the
arithmetic on the series (fp64 normalise -> fp32) runs in the HIP kernel.  No CPU fallback: a non-HIP device raises.
"""
import numpy as np
import torch

from . import _lib, ops


def _fill_na(data):
    """`fillna(ffill).fillna(bfill)` of forecast_dataloader.py:49 (load-time, host)."""
    data = np.array(data, dtype=np.float64, copy=True)
    if data.ndim == 1:
        data = data[:, None]
    if not np.isnan(data).any():
        return data
    T = data.shape[0]
    idx = np.where(~np.isnan(data), np.arange(T)[:, None], -1)
    np.maximum.accumulate(idx, axis=0, out=idx)                           # last valid row at or before t
    filled = np.take_along_axis(data, np.maximum(idx, 0), axis=0)
    filled[idx < 0] = np.nan
    idx = np.where(~np.isnan(filled), np.arange(T)[:, None], T)
    idx = np.minimum.accumulate(idx[::-1], axis=0)[::-1]                  # first valid row at or after t
    out = np.take_along_axis(filled, np.minimum(idx, T - 1), axis=0)
    out[idx >= T] = np.nan
    return out


def _stat_arrays(data, normalize_method, norm_statistic):
    """(sub, div, clip01, statistic) of normalized() (forecast_dataloader.py:7-22), float64 host arrays.

    The reference cannot take list statistics for min_max (`list - list`, :11); arrays and lists are both accepted
    here."""
    if normalize_method == "min_max":
        if not norm_statistic:
            norm_statistic = dict(max=np.max(data, axis=0), min=np.min(data, axis=0))
        lo = np.asarray(norm_statistic["min"], dtype=np.float64)
        return lo, np.asarray(norm_statistic["max"], dtype=np.float64) - lo + 1e-5, True, norm_statistic
    if normalize_method == "z_score":
        if not norm_statistic:
            norm_statistic = dict(mean=np.mean(data, axis=0), std=np.std(data, axis=0))
        std = [1 if i == 0 else i for i in norm_statistic["std"]]
        norm_statistic["std"] = std                                       # the reference mutates the dict (:21)
        return (np.asarray(norm_statistic["mean"], dtype=np.float64), np.asarray(std, dtype=np.float64), False,
                norm_statistic)
    raise ValueError(f"unknown normalize_method {normalize_method!r}")


def _device(device):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.StemGNNHipError(f"stemgnn_amd data path runs only on a HIP device, got {dev} (no CPU fallback)")
    return dev


def normalized(data, normalize_method, norm_statistic=None, device="cuda"):
    """forecast_dataloader.py:7-22 -> (float32 device tensor [T,N], norm_statistic).  Unknown / empty method: cast only."""
    dev = _device(device)
    data = np.asarray(data, dtype=np.float64)
    N = data.shape[1]
    if normalize_method in ("min_max", "z_score"):
        sub, div, clip, norm_statistic = _stat_arrays(data, normalize_method, norm_statistic)
    else:
        sub, div, clip = np.zeros(N), np.ones(N), False
    raw = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    out = ops.normalize_series(raw, torch.from_numpy(np.ascontiguousarray(sub)).to(dev),
                               torch.from_numpy(np.ascontiguousarray(div)).to(dev), clip)
    return out, norm_statistic


def denorm_coefficients(normalize_method, norm_statistic, device):
    """(mul, add) float64 device vectors with de_normalized(v) = v * mul + add (forecast_dataloader.py:25-38)."""
    if normalize_method == "min_max":
        lo = np.asarray(norm_statistic["min"], dtype=np.float64)
        mul, add = np.asarray(norm_statistic["max"], dtype=np.float64) - lo + 1e-8, lo
    elif normalize_method == "z_score":
        mul = np.asarray([1 if i == 0 else i for i in norm_statistic["std"]], dtype=np.float64)
        add = np.asarray(norm_statistic["mean"], dtype=np.float64)
    else:
        return None, None
    return torch.from_numpy(mul).to(device), torch.from_numpy(add).to(device)


def de_normalized(data, normalize_method, norm_statistic):
    """forecast_dataloader.py:25-38 on a device tensor [..., N]; float64 result as in the reference."""
    if not norm_statistic:
        raise ValueError("de_normalized needs the statistics the data was normalised with")
    mul, add = denorm_coefficients(normalize_method, norm_statistic, data.device)
    if mul is None:
        return data
    return data.double() * mul + add


class ForecastDataset(torch.utils.data.Dataset):
    """forecast_dataloader.py:41-73 with `.data` resident on the GPU ([T,N] fp32 -- what `__getitem__`'s
    `.type(torch.float)` would produce row by row)."""

    def __init__(self, df, window_size, horizon, normalize_method=None, norm_statistic=None, interval=1,
                 device="cuda"):
        self.window_size = window_size
        self.interval = interval
        self.horizon = horizon
        self.normalize_method = normalize_method
        self.norm_statistic = norm_statistic
        self.device = _device(device)
        host = _fill_na(np.asarray(df, dtype=np.float64))
        self.df_length = len(host)
        self.x_end_idx = self.get_x_end_idx()
        self.data, _ = normalized(host, normalize_method, norm_statistic, device=self.device)
        self.hi_all = torch.tensor(self.x_end_idx, dtype=torch.int64, device=self.device)

    def get_x_end_idx(self):
        x_index_set = range(self.window_size, self.df_length - self.horizon + 1)
        return [x_index_set[j * self.interval] for j in range((len(x_index_set)) // self.interval)]

    def __len__(self):
        return len(self.x_end_idx)

    def gather(self, indices, x=None, y=None):
        """indices: int64 device tensor (or list) of dataset indices -> (x [B,W,N], y [B,H,N])."""
        if not torch.is_tensor(indices):
            indices = torch.tensor(list(indices), dtype=torch.int64, device=self.device)
        hi = self.hi_all.index_select(0, indices)
        return ops.window_gather(self.data, hi, self.window_size, self.horizon, x, y)

    def __getitem__(self, index):
        if not -len(self) <= index < len(self):
            raise IndexError(index)
        x, y = self.gather([index % len(self)])
        return x[0], y[0]


def epoch_order(n, shuffle, generator=None):
    """Dataset indices of one pass, consuming the torch RNG exactly like ``iter(DataLoader(ds, shuffle=...))`` does
    with num_workers=0 (torch.utils.data: the iterator draws its base seed, then RandomSampler seeds a private
    generator from the global RNG and takes ``randperm(n)``).  Host logic, as in the reference."""
    torch.empty((), dtype=torch.int64).random_(generator=generator)       # _BaseDataLoaderIter._base_seed
    if not shuffle:
        return list(range(n))
    if generator is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        generator = torch.Generator()
        generator.manual_seed(seed)
    return torch.randperm(n, generator=generator).tolist()


class WindowLoader:
    """Stands in for ``torch_data.DataLoader(dataset, batch_size, shuffle, drop_last, num_workers=0)``
    (models/handler.py:136-138) over a device-resident ForecastDataset: yields (x [B,W,N], y [B,H,N]) device batches."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.generator = generator
        self.last_order = None

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def index_batches(self):
        """One pass as int64 device index tensors (one H2D copy of the whole permutation per epoch)."""
        order = epoch_order(len(self.dataset), self.shuffle, self.generator)
        self.last_order = order
        dev_order = torch.tensor(order, dtype=torch.int64, device=self.dataset.device)
        for s in range(0, len(order), self.batch_size):
            chunk = dev_order[s:s + self.batch_size]
            if self.drop_last and chunk.numel() < self.batch_size:
                return
            yield chunk

    def __iter__(self):
        for chunk in self.index_batches():
            yield self.dataset.gather(chunk)
