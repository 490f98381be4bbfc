"""Generate golden vectors from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports /root/reference/models/base_model.py under oracle/ref_shim.py, loads the
deterministic weights of oracle.stemgnn_oracle.det_state_dict, runs forward + MSE
backward on deterministic inputs and writes tests/golden/<case>.npz:
  x, y, forecast, attention, loss, mul_L, and per-parameter gradients
  (full for small cases; for the real-shape case the big GLU gradients are stored
  as their first 64 elements + L2 norm + sum to keep the fixture small).
The reference has no tests or golden vectors of its own, so
these -- outputs of the reference itself -- are what pins the oracle.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.detrand import det_normalish, det_uniform  # noqa: E402
from oracle.ref_shim import load_reference_model_module  # noqa: E402
from oracle.stemgnn_oracle import det_state_dict  # noqa: E402

# name: (N, W, multi, H, B, mode, full_grads)
CASES = {
    "tiny_eval_h1":   dict(N=9,  W=4,  multi=2, H=1, B=3, mode="eval",  full=True),
    "small_train_p0": dict(N=12, W=6,  multi=3, H=3, B=5, mode="train", full=True),   # even Wm=18
    "odd_wm_train":   dict(N=7,  W=5,  multi=3, H=2, B=2, mode="train", full=True),   # odd  Wm=15
    "small_dropmask": dict(N=10, W=4,  multi=2, H=2, B=4, mode="mask",  full=True),   # train mode, fixed dropout mask
    "pems_shape_n20": dict(N=20, W=12, multi=5, H=3, B=4, mode="train", full=False),  # real W/multi (C=240)
}


def run_case(name, c, ref):
    torch.manual_seed(0)
    N, W, m, H, B = c["N"], c["W"], c["multi"], c["H"], c["B"]
    p = 0.5 if c["mode"] == "mask" else 0.0
    model = ref.Model(N, 2, W, m, horizon=H, dropout_rate=p)
    sd = det_state_dict(N, W, m, H, seed=hash_seed(name))
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    x = torch.from_numpy(det_normalish((B, W, N), 7 + hash_seed(name)))
    y = torch.from_numpy(det_normalish((B, H, N), 11 + hash_seed(name)))
    out = {"x": x.numpy(), "y": y.numpy()}
    if c["mode"] == "eval":
        model.eval()
    else:
        model.train()
    if c["mode"] == "mask":
        mask = (det_uniform((B, N, N), 13 + hash_seed(name), 0.0, 1.0) >= p).astype(np.float32)
        out["drop_mask"] = mask
        tm = torch.from_numpy(mask)

        class FixedMask(torch.nn.Module):                  # stands in for nn.Dropout's Bernoulli draw (:161)
            def forward(self, t):
                return t * tm / (1.0 - p)

        model.dropout = FixedMask()
    model.zero_grad()
    mul_L, _ = model.latent_correlation_layer(x)
    forecast, att = model(x)
    loss = torch.nn.functional.mse_loss(forecast, y)
    loss.backward()
    out.update(forecast=forecast.detach().numpy(), attention=att.detach().numpy(),
               loss=np.float64(loss.item()), mul_L=mul_L.detach().numpy())
    for k, prm in model.named_parameters():
        g = prm.grad
        if g is None:
            out["gradnone." + k] = np.zeros(1, np.float32)
            continue
        g = g.detach().numpy()
        if c["full"] or g.size <= 4096:
            out["grad." + k] = g
        else:
            flat = g.reshape(-1).astype(np.float64)
            out["gradhead." + k] = g.reshape(-1)[:64].copy()
            out["gradstat." + k] = np.array([np.sqrt((flat ** 2).sum()), flat.sum(), np.abs(flat).max()])
    cfg = np.array([N, W, m, H, B, {"eval": 0, "train": 1, "mask": 2}[c["mode"]]], np.int64)
    out["cfg"] = cfg
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: loss={loss.item():.6f} |forecast|max={forecast.abs().max():.4f} "
          f"size={os.path.getsize(os.path.join(HERE, name + '.npz')) / 1024:.1f} KiB")


def hash_seed(name):
    return sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 9973


if __name__ == "__main__":
    ref = load_reference_model_module()
    for name, c in CASES.items():
        run_case(name, c, ref)
