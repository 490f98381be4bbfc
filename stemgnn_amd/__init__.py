"""stemgnn_amd -- MI355X-native (gfx950) implementation of StemGNN's spectral hot path.

The product path is hand-written HIP behind a C ABI (``include/stemgnn_hip.h`` ->
``stemgnn_amd/libstemgnn_hip.so``); this package is the host-side mirror of the reference's
``models.base_model`` interface (``Model``, ``StockBlockLayer``, ``GLU``).  There is no CPU
fallback: calling the model without the HIP library or on a non-GPU tensor raises.
"""
from .base_model import GLU, Model, StockBlockLayer  # noqa: F401

__all__ = ["Model", "StockBlockLayer", "GLU"]
__version__ = "0.1.0"
