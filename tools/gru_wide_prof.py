import os, sys
sys.path.insert(0, os.getcwd())
import torch
from stemgnn_amd.ops import GruFront, check_gru_status
dev = torch.device("cuda")
for B, W, S in ((8, 12, 1024), (16, 48, 2048)):
    g = torch.nn.GRU(W, S).to(dev)
    x = torch.randn(B, W, S, device=dev)
    ps = [g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0]
    with torch.no_grad():
        GruFront.apply(x, *ps)
    torch.cuda.synchronize()
    check_gru_status(dev)
    print("== done", B, S, flush=True)
