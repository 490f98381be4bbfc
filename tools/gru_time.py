"""Time the GRU stage alone (C ABI), fwd and bwd, for a few STEMGNN_GRU_NW settings (subprocess per setting)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def one():
    import torch
    from stemgnn_amd.ops import GruFront
    B, W, S = 32, 12, int(os.environ.get("GRU_N", "228"))
    dev = torch.device("cuda")
    g = torch.nn.GRU(W, S).to(dev)
    x = torch.randn(B, W, S, device=dev)
    ps = [g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0]
    dh = torch.randn(S, B, S, device=dev)
    def fb():
        for p in ps: p.grad = None
        h = GruFront.apply(x, *ps); h.backward(dh)
    for _ in range(3): fb()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.no_grad():
        e[0].record()
        for _ in range(10): GruFront.apply(x, *ps)
        e[1].record()
    for _ in range(10): fb()
    e[2].record(); torch.cuda.synchronize()
    f = e[0].elapsed_time(e[1]) / 10; t = e[1].elapsed_time(e[2]) / 10
    print(f"NW={os.environ.get('STEMGNN_GRU_NW','auto'):>4} N={S}: fwd {f*1e3:8.1f} us  fwd+bwd {t*1e3:8.1f} us", flush=True)
if len(sys.argv) > 1: one()
else:
    for n in ("228", "140", "358"):
        for nw in ("4", "8", "16"):
            env = dict(os.environ, STEMGNN_GRU_NW=nw, GRU_N=n)
            r = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True, timeout=300)
            print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
