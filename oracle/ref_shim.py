"""TEST INFRASTRUCTURE ONLY -- import the real reference (read-only, /root/reference) on torch 2.x.

Only usable in the build container (``/root/reference`` does not exist on the GPU box).
Applies the 4-item compat shim of SURVEY.md section 8c by monkey-patching *before* import:
  1. torch.rfft(x, 1, onesided=False)  := view_as_real(fft.fft(x, dim=-1))        (models/base_model.py:49)
  2. torch.irfft(z, 1, onesided=False) := fft.irfft(view_as_complex(z)[..., :n//2+1], n=n)  (models/base_model.py:58)
  3. np.float = float                                                             (models/handler.py:50)
  4. torch.load(..., weights_only=False)                                          (models/handler.py:37)
Nothing is written to /root/reference (sys.dont_write_bytecode).
"""
import importlib
import os
import sys

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("STEMGNN_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "base_model.py"))


def _install_shims():
    if not hasattr(torch, "rfft") or getattr(torch.rfft, "_stemgnn_shim", False) is False:
        def rfft(x, signal_ndim, normalized=False, onesided=True):
            assert signal_ndim == 1 and not normalized and not onesided
            return torch.view_as_real(torch.fft.fft(x, dim=-1))

        def irfft(z, signal_ndim, normalized=False, onesided=True, signal_sizes=None):
            assert signal_ndim == 1 and not normalized and not onesided
            n = z.shape[-2]
            zc = torch.view_as_complex(z.contiguous())
            return torch.fft.irfft(zc[..., : n // 2 + 1], n=n, dim=-1)

        rfft._stemgnn_shim = True
        irfft._stemgnn_shim = True
        torch.rfft = rfft
        torch.irfft = irfft
    if not hasattr(np, "float"):
        np.float = float


def load_reference_model_module():
    """Return the reference's ``models.base_model`` module (imported from REFERENCE_ROOT)."""
    if not reference_available():
        raise FileNotFoundError(f"reference not mounted at {REFERENCE_ROOT}")
    _install_shims()
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "_stemgnn_reference_base_model", os.path.join(REFERENCE_ROOT, "models", "base_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_packages():
    """Import the reference's ``data_loader.forecast_dataloader``, ``utils.math_utils`` and ``models.handler``
    (as the packages they are, REFERENCE_ROOT prepended to sys.path) under the compat shims.  Returns the 3 modules."""
    if not reference_available():
        raise FileNotFoundError(f"reference not mounted at {REFERENCE_ROOT}")
    _install_shims()
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings
    warnings.filterwarnings("ignore", category=FutureWarning)
    fd = importlib.import_module("data_loader.forecast_dataloader")
    mu = importlib.import_module("utils.math_utils")
    hd = importlib.import_module("models.handler")
    return fd, mu, hd
