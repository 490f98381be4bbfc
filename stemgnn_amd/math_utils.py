"""Device mirror of the reference's ``utils/math_utils.py``: MAPE / MAE / RMSE / evaluate on
[count, time_step, node] tensors that stay on the GPU; one fused fp64 kernel pass (`stemgnn_eval_metrics`) yields
every axis variant, and only the few result numbers come back to the host.

Quirks kept (utils/math_utils.py:32-33): MAPE adds 1e-5 to each ratio and clips at 5; a 0/0 stays NaN.
"""
import numpy as np
import torch

from . import ops


def _as_f32(t, like=None):
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
        if like is not None:
            t = t.to(like.device)
    return t.float() if t.dtype != torch.float32 else t


class Scores:
    """All axis variants of (MAPE, MAE, RMSE) from one kernel pass."""

    def __init__(self, y, y_hat, mul=None, add=None):
        y_hat = _as_f32(y_hat)
        y = _as_f32(y, like=y_hat)
        C, H, N = y.shape
        v = ops.eval_metrics(y, y_hat, mul, add).cpu().numpy()
        self.overall = v[:3]
        o = 3
        self.by_node = v[o:o + 3 * N].reshape(3, N); o += 3 * N
        self.by_step = v[o:o + 3 * H].reshape(3, H); o += 3 * H
        self.by_step_node = v[o:o + 3 * H * N].reshape(3, H, N)

    def get(self, by_step=False, by_node=False):
        if by_step and by_node:
            m = self.by_step_node
        elif by_step:
            m = self.by_step
        elif by_node:
            m = self.by_node
        else:
            return tuple(np.float64(x) for x in self.overall)
        return m[0].copy(), m[1].copy(), m[2].copy()


def evaluate(y, y_hat, by_step=False, by_node=False):
    """utils/math_utils.py:59-74.  y: ground truth, y_hat: prediction, both [count, time_step, node] on the GPU."""
    return Scores(y, y_hat).get(by_step, by_node)


_AXES = {None: (False, False), 0: (True, True), (0, 2): (True, False), (0, 1): (False, True)}


def _one(which, v, v_, axis):
    if axis not in _AXES:
        raise ValueError(f"axis {axis!r}: the device metrics cover the variants evaluate() uses: {list(_AXES)}")
    return Scores(v, v_).get(*_AXES[axis])[which]


def MAPE(v, v_, axis=None):
    return _one(0, v, v_, axis)


def MAE(v, v_, axis=None):
    return _one(1, v, v_, axis)


def RMSE(v, v_, axis=None):
    return _one(2, v, v_, axis)
