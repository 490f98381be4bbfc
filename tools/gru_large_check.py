"""GRU forward/backward accuracy at large hidden sizes against torch CPU fp64 (and torch CPU fp32 for scale).
GPU tool: python tools/gru_large_check.py [B Hd W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stemgnn_amd.ops import GruFront  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, Hd, W = a if len(a) == 3 else (2, 2048, 48)
    torch.manual_seed(0)
    gru = torch.nn.GRU(W, Hd)
    x = torch.randn(B, W, Hd)
    dh = torch.randn(Hd, B, Hd) * 1e-3
    outs = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        g = torch.nn.GRU(W, Hd).to(dt)
        g.load_state_dict({k: v.to(dt) for k, v in gru.state_dict().items()})
        h, _ = g(x.permute(2, 0, 1).contiguous().to(dt))
        (h * dh.to(dt)).sum().backward()
        outs[name] = (h.detach(), {k: p.grad.detach() for k, p in g.named_parameters()})
    dev = torch.device("cuda")
    prm = [p.detach().to(dev).requires_grad_(True) for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
    h = GruFront.apply(x.to(dev), *prm)
    (h * dh.to(dev)).sum().backward()
    torch.cuda.synchronize()
    print(f"B={B} Hd={Hd} W={W}")
    print(f"  h        hip vs f64 {rel(h.detach(), outs['f64'][0]):.2e}   torch f32 vs f64 {rel(outs['f32'][0], outs['f64'][0]):.2e}")
    for p, k in zip(prm, ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")):
        print(f"  d{k:14s} hip vs f64 {rel(p.grad, outs['f64'][1][k]):.2e}   torch f32 vs f64 {rel(outs['f32'][1][k], outs['f64'][1][k]):.2e}")
    hs = h.detach().double().cpu()
    err = (hs - outs["f64"][0]).abs()
    print("  max |h err| by step (first, mid, last):", [f"{float(err[i].max()):.1e}" for i in (0, Hd // 2, Hd - 1)])


if __name__ == "__main__":
    main()
