"""Debug helper (GPU): per-tensor error table of the HIP path vs the CPU oracle for one config.
usage: python tools/gpu_check.py N W multi H B [drop_p]"""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import stemgnn_oracle as O
from stemgnn_amd import Model
from stemgnn_amd import ops


def rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    N, W, multi, H, B = (int(v) for v in sys.argv[1:6])
    p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
    dev = torch.device("cuda:0")
    sd = O.det_state_dict(N, W, multi, H, seed=3)
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=p)
    model.load_state_dict(sd)
    model.to(dev).train()
    torch.manual_seed(0)
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)
    kw = {}
    if p > 0:
        model.set_dropout_seed(1234, 5)
        seed = model._seed.clone()
        mask = ops.dropout_mask(p, seed, B, N).cpu()
        print("keep fraction", float(mask.mean()))
        kw = dict(drop_mask=mask, drop_p=p)
    t0 = time.time()
    fsum, att, mul_L = model.hot_path(x.to(dev))
    forecast, _ = None, None
    y_hat = model.fc(fsum)
    forecast = y_hat.permute(0, 2, 1).contiguous() if y_hat.size(-1) != 1 else y_hat.unsqueeze(1).squeeze(-1)
    loss = torch.nn.functional.mse_loss(forecast, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    print(f"gpu fwd+bwd wall {time.time() - t0:.3f}s (first call)")
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    gru_out = O.gru_front(x, leaves)
    o_fsum, o_att, o_mulL = O.hot_path(gru_out, x, leaves, **kw)
    o_forecast, _ = O.model_forward(x, leaves, **kw)
    o_loss = torch.nn.functional.mse_loss(o_forecast, y)
    o_loss.backward()
    print(f"loss gpu {loss.item():.7f} oracle {o_loss.item():.7f}")
    rows = [("attention", rel(att, o_att)), ("mul_L[1]", rel(mul_L[1], o_mulL[1])), ("mul_L[2]", rel(mul_L[2], o_mulL[2])),
            ("mul_L[3]", rel(mul_L[3], o_mulL[3])), ("fsum", rel(fsum, o_fsum)), ("forecast", rel(forecast, o_forecast))]
    for k, prm in model.named_parameters():
        og = leaves[k].grad
        if og is None:
            rows.append(("grad." + k, -1.0 if prm.grad is None else float(prm.grad.abs().max())))
        elif prm.grad is None:
            rows.append(("grad." + k + " MISSING", 9.9))
        else:
            rows.append(("grad." + k, rel(prm.grad, og)))
    bad = 0
    for k, e in rows:
        flag = "" if e < 1e-4 else "  <-- FAIL"
        bad += e >= 1e-4
        print(f"{k:55s} {e:10.3e}{flag}")
    print("FAILS:", bad)


if __name__ == "__main__":
    main()
