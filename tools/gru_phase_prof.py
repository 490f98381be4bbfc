"""Phase cycle counters of the cluster GRU (needs a -DGRU_PROF build of the library: STEMGNN_HIP_LIB=build/ab/lib_prof.so).
Runs the PEMS07-shape recurrence forward + backward a few times; the kernels print their per-step phase averages."""
import sys
import torch

sys.path.insert(0, ".")
from stemgnn_amd.ops import GruFront, check_gru_status

B, S, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 228, 12)
torch.manual_seed(0)
gru = torch.nn.GRU(W, S)
x = torch.randn(B, W, S).cuda()
dh = torch.randn(S, B, S).cuda()
params = [p.detach().clone().cuda().requires_grad_(True) for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
for it in range(3):
    print(f"--- run {it}", flush=True)
    h = GruFront.apply(x, *params)
    h.backward(dh)
    torch.cuda.synchronize()
check_gru_status(torch.device("cuda:0"))
