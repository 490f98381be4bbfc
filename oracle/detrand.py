"""TEST INFRASTRUCTURE ONLY -- version-independent deterministic pseudo-random tensors.

Golden fixtures (tests/golden/) store inputs/outputs of the *reference* model but
not its (multi-MB) weights; both the generator script and the tests rebuild the
weights from this splitmix64 stream, which depends on nothing but integer
arithmetic (no numpy/torch RNG version drift).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array of `shape`, i.i.d.-looking uniform in [lo, hi), a pure function of (shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        bits = _splitmix64(_splitmix64(idx))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_normalish(shape, seed):
    """zero-mean, unit-variance (sum of 4 uniforms) float32 array; deterministic."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += det_uniform(shape, seed * 4 + k + 1000003).astype(np.float64)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)
