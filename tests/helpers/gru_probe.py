"""GRU front alone (forward + backward through the C ABI) on fixed inputs: prints one JSON line with SHA-256 digests of
the hidden states and of every gradient, and the time per forward / backward.  Run as a subprocess by
tests/test_hip_gru_eigh.py with STEMGNN_HIP_LIB pointing at a build variant (the library is chosen at import time)."""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from stemgnn_amd import _lib, ops

    B, W, S = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 12, 228)
    torch.manual_seed(0)
    gru = torch.nn.GRU(W, S)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, W, S, generator=g).cuda()
    dh = (torch.randn(S, B, S, generator=g) * 0.01).cuda()
    gru = gru.cuda()
    ps = [gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0]

    def fwd_bwd():
        for p in ps:
            p.grad = None
        h = ops.GruFront.apply(x, *ps)
        h.backward(dh)
        return h
    h = fwd_bwd()
    torch.cuda.synchronize()
    ops.check_gru_status(x.device)

    def sha(t):
        return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()
    out = {"lib": os.path.basename(_lib.LIB_PATH), "h": sha(h), "grads": [sha(p.grad) for p in ps]}
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.no_grad():
        e[0].record()
        for _ in range(10):
            ops.GruFront.apply(x, *ps)
        e[1].record()
    for _ in range(10):
        fwd_bwd()
    e[2].record()
    torch.cuda.synchronize()
    ops.check_gru_status(x.device)
    out["fwd_us"] = e[0].elapsed_time(e[1]) * 100.0
    out["bwd_us"] = e[1].elapsed_time(e[2]) * 100.0 - out["fwd_us"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
