"""CPU experiment (oracle only): the LeakyReLU kink of the attention logits (models/base_model.py:159) at the configs[4]
shape.  Lists the smallest |key_i + query_j|, the fp32 rounding of key/query, and how much the ~1e-8 gradients of
weight_key / weight_query / the GRU move (fp64 arithmetic) when the sign decision of the k elements closest to the kink is
flipped -- what any two fp32 implementations with different summation orders may legitimately disagree on.
    python tools/kink_probe.py [N W multi H B]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stemgnn_oracle as O  # noqa: E402


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max())


def main():
    a = [int(v) for v in sys.argv[1:]]
    N, W, multi, H, B = a if len(a) == 5 else (2048, 48, 5, 12, 2)
    torch.set_num_threads(32)
    sd32 = O.det_state_dict(N, W, multi, H, seed=N)
    sd = {k: v.double() for k, v in sd32.items()}
    torch.manual_seed(N)
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)
    keys = ["weight_key", "weight_query", "GRU.weight_ih_l0", "GRU.weight_hh_l0", "stock_block.0.weight"]
    _, _, _, ref = O.loss_and_grads(x.double(), y.double(), sd)
    pre = {}
    for name, s_, xx in (("f64", sd, x.double()), ("f32", sd32, x)):
        inp = O.gru_front(xx, s_).permute(0, 2, 1)
        pre[name] = torch.matmul(inp, s_["weight_key"]) + torch.matmul(inp, s_["weight_query"]).transpose(1, 2)
    a64 = pre["f64"].abs().flatten()
    vals, idx = torch.topk(a64, 8, largest=False)
    print("smallest |key_i + query_j| (fp64):", [f"{v:.1e}" for v in vals.tolist()])
    print("max |fp32 - fp64| of the logits:", f"{float((pre['f32'].double() - pre['f64']).abs().max()):.1e}",
          " sign decisions that differ between torch fp32 and fp64:", int(((pre["f32"] > 0) != (pre["f64"] > 0)).sum()))
    real = F.leaky_relu
    for k in (1, 2, 4):
        flip = torch.zeros(a64.numel(), dtype=torch.bool)
        flip[idx[:k]] = True
        flip = flip.view_as(pre["f64"])

        def patched(t, alpha=0.01, _flip=flip):
            if t.shape == _flip.shape and t.dtype == torch.float64:
                return torch.where((t > 0) ^ _flip, t, alpha * t)
            return real(t, alpha)

        O.F.leaky_relu = patched
        try:
            _, _, _, got = O.loss_and_grads(x.double(), y.double(), sd)
        finally:
            O.F.leaky_relu = real
        print(f"flip the {k} closest:", {kk: f"{relerr(got[kk], ref[kk]):.1e}" for kk in keys})


if __name__ == "__main__":
    main()
