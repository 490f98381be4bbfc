"""CPU experiment (oracle only): how much do d loss/d weight_key, weight_query and the GRU gradients move when an
intermediate of the backward pass carries fp32-sized relative noise?  fp64 oracle at the configs[4] shape; a hook
multiplies the gradient arriving at a chosen tensor by (1 + eps * N(0,1)) elementwise.
    python tools/conditioning_probe.py [N W multi H B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stemgnn_oracle as O  # noqa: E402


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max())


def main():
    a = [int(v) for v in sys.argv[1:]]
    N, W, multi, H, B = a if len(a) == 5 else (2048, 48, 5, 12, 2)
    torch.set_num_threads(32)
    sd = {k: v.double() for k, v in O.det_state_dict(N, W, multi, H, seed=N).items()}
    torch.manual_seed(N)
    x, y = torch.randn(B, W, N).double(), torch.randn(B, H, N).double()
    _, _, _, ref = O.loss_and_grads(x, y, sd)
    keys = ["weight_key", "weight_query", "GRU.weight_hh_l0", "GRU.weight_ih_l0", "stock_block.0.weight"]
    print("magnitudes:", {k: f"{float(ref[k].abs().max()):.2e}" for k in keys})
    real = {"lap": O.laplacian_from_attention, "cheb": O.cheb_polynomial, "att": O.self_graph_attention}

    def run(tag, eps, where):
        g = torch.Generator().manual_seed(1)

        def noisy(t):
            t.register_hook(lambda gr: gr * (1 + eps * torch.randn(gr.shape, generator=g, dtype=gr.dtype)))
            return t

        if where == "dL":        # gradient w.r.t. the Laplacian L (input of cheb_polynomial)
            O.cheb_polynomial = lambda L: real["cheb"](noisy(L))
        elif where == "dmul_L":  # gradient w.r.t. the Chebyshev stack
            O.cheb_polynomial = lambda L: noisy(real["cheb"](L))
        elif where == "dA":      # gradient w.r.t. the batch-mean attention
            O.laplacian_from_attention = lambda att: real["lap"](noisy(att))
        try:
            _, _, _, got = O.loss_and_grads(x, y, sd)
        finally:
            O.cheb_polynomial, O.laplacian_from_attention = real["cheb"], real["lap"]
        print(f"{tag:28s}", {k: f"{relerr(got[k], ref[k]):.1e}" for k in keys})

    for where in ("dmul_L", "dL", "dA"):
        for eps in (1e-7, 1e-6):
            run(f"noise {eps:g} on {where}", eps, where)


if __name__ == "__main__":
    main()
