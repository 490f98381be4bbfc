"""Bisect hipGraph capture problems: run each piece in a subprocess (a segfault only kills that piece)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PIECES = ["gru_fwd", "gru_fwd_bwd", "hot_fwd", "model_fwd_bwd", "model_fwd_bwd_miopen", "model_step", "model_step_nobucket"]


def run_piece(name):
    import torch

    from stemgnn_amd import Model
    from stemgnn_amd.distributed import FlatGradBucket
    from stemgnn_amd.ops import GruFront

    dev = torch.device("cuda")
    N, W, multi, H, B = 228, 12, 5, 3, 32
    if name.endswith("miopen"):
        os.environ["STEMGNN_GRU"] = "miopen"
    torch.manual_seed(0)
    model = Model(N, 2, W, multi, horizon=H).to(dev).train()
    x, y = torch.randn(B, W, N, device=dev), torch.randn(B, H, N, device=dev)
    g = model.GRU
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-4, eps=1e-8, capturable=True, foreach=True)
    bucket = FlatGradBucket(model.parameters()) if name == "model_step" else None

    def fn():
        if name == "gru_fwd":
            with torch.no_grad():
                GruFront.apply(x, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        elif name == "gru_fwd_bwd":
            h = GruFront.apply(x, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
            h.sum().backward()
        elif name == "hot_fwd":
            with torch.no_grad():
                model(x)
        elif name.startswith("model_fwd_bwd"):
            f, _ = model(x)
            torch.nn.functional.mse_loss(f, y).backward()
        else:
            if bucket is not None:
                bucket.zero()
            else:
                opt.zero_grad(set_to_none=True)
            f, _ = model(x)
            torch.nn.functional.mse_loss(f, y).backward()
            opt.step()

    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    torch.cuda.synchronize()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    print("PIECE_OK", name, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_piece(sys.argv[1])
    else:
        for p in PIECES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, p], capture_output=True, text=True,
                               timeout=180)
            ok = "PIECE_OK" in r.stdout
            print(f"{p:28s} rc={r.returncode} {'OK' if ok else 'FAIL'}", flush=True)
            if not ok:
                print("   ", "\n    ".join(r.stderr.strip().splitlines()[-8:]), flush=True)
