"""Stage-level accuracy probe of stemgnn_attn_laplacian_fwd/_bwd against an fp64 torch reference (no GRU involved).
usage: python tools/attn_stage_check.py N B [mode]   mode: rand | struct (dL with large row/column-constant parts)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import stemgnn_oracle as O
from stemgnn_amd import _lib

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))

N, B = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "rand"
torch.manual_seed(0)
h = torch.tanh(torch.randn(N, B, N))                       # like a GRU output
wk = (torch.rand(N, 1) - 0.5) * 0.2
wq = (torch.rand(N, 1) - 0.5) * 0.2
dL = torch.randn(N, N) * 1e-3
if mode == "struct":
    dL = dL + torch.randn(N, 1) + torch.randn(1, N)
lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
hg, wkg, wqg, dLg = h.to(dev), wk.to(dev), wq.to(dev), dL.to(dev)
saved = torch.empty(lib.stemgnn_attn_saved_floats(B, N), device=dev)
att = torch.empty(N, N, device=dev); mulL = torch.empty(4, N, N, device=dev)
_lib.check(lib.stemgnn_attn_laplacian_fwd(hg.data_ptr(), wkg.data_ptr(), wqg.data_ptr(), 0.2, 0.0, 1, None, B, N,
                                          saved.data_ptr(), att.data_ptr(), mulL.data_ptr(), 3, st), "fwd")
nch = 4
scr = torch.empty(lib.stemgnn_attn_scratch_floats(B, N, nch), device=dev)
dh = torch.empty_like(hg); dwk = torch.empty_like(wkg); dwq = torch.empty_like(wqg)
_lib.check(lib.stemgnn_attn_laplacian_bwd(dLg.data_ptr(), hg.data_ptr(), wkg.data_ptr(), wqg.data_ptr(), 0.2, 0.0, 1, None,
                                          B, N, saved.data_ptr(), scr.data_ptr(), nch, dh.data_ptr(), dwk.data_ptr(),
                                          dwq.data_ptr(), 3, st), "bwd")
torch.cuda.synchronize()
res = {}
for dt in (torch.float64, torch.float32):
    hh = h.to(dt).requires_grad_(True); k = wk.to(dt).requires_grad_(True); q = wq.to(dt).requires_grad_(True)
    a = O.self_graph_attention(hh.permute(1, 0, 2), k, q)          # [B, N_seq, N_hid] view of h[s,b,i]
    L, A_s = O.laplacian_from_attention(a)
    (L * dL.to(dt)).sum().backward()
    res[dt] = (L.detach(), A_s.detach(), hh.grad, k.grad, q.grad)
t, o = res[torch.float64], res[torch.float32]
for name, mine, i in (("L", mulL[1], 0), ("attention", att, 1), ("dh", dh, 2), ("dwk", dwk, 3), ("dwq", dwq, 4)):
    print(f"{name:10s} hip-vs-fp64 {rel(mine, t[i]):9.2e}   torch32-vs-fp64 {rel(o[i], t[i]):9.2e}   |truth|max {float(t[i].abs().max()):.3e}")
