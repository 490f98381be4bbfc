"""Shared helpers for the test-suite (test infrastructure)."""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def hash_seed(name):
    return sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 9973


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    N, W, m, H, B, mode = (int(v) for v in z["cfg"])
    cfg = dict(N=N, W=W, multi=m, H=H, B=B, mode={0: "eval", 1: "train", 2: "mask"}[mode])
    return z, cfg


def relerr(got, ref):
    """norm-relative error used throughout: max|got-ref| / max|ref|."""
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    den = ref.abs().max().item()
    num = (got - ref).abs().max().item()
    return num / den if den > 0 else num


def synthetic_series(T, N, seed):
    """sin(2 pi t / p_n + phi_n) * a_n + c_n + noise (SURVEY 8d shape), float64, from the deterministic generators of
    oracle/detrand.py (so a fixture can name a seed instead of carrying the series)."""
    from oracle.detrand import det_normalish, det_uniform
    t = np.arange(T, dtype=np.float64)[:, None]
    p = det_uniform((N,), seed, 6.0, 30.0).astype(np.float64)
    phi = det_uniform((N,), seed + 1, 0.0, 6.28).astype(np.float64)
    a = det_uniform((N,), seed + 2, 0.5, 3.0).astype(np.float64)
    c = det_uniform((N,), seed + 3, -2.0, 8.0).astype(np.float64)
    return np.sin(2 * np.pi * t / p + phi) * a + c + 0.1 * det_normalish((T, N), seed + 4).astype(np.float64)
