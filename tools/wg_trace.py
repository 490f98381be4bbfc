"""Placement / span trace of ONE launch of the fused weight-gradient kernel (debug library, STEMGNN_WG_DEBUG bit 16).
usage (GPU box):  STEMGNN_HIP_LIB=.../libstemgnn_hip_dbg.so STEMGNN_WG_DEBUG=16 python tools/wg_trace.py > trace.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stemgnn_amd import _lib, ops  # noqa: E402

B, N, W, multi = 32, 228, 12, 5
lib = _lib.load()
dev = torch.device("cuda")
nsplit = ops._NSPLIT
packed = torch.randn(lib.stemgnn_packed_floats(W, multi), device=dev) * 0.05
saved = torch.randn(lib.stemgnn_saved_floats(B, N, W, multi), device=dev)
scratch = torch.randn(lib.stemgnn_scratch_floats(B, N, W, multi), device=dev) * 0.1
gradpart = torch.empty(lib.stemgnn_gradpart_floats(W, multi, nsplit), device=dev)
st = torch.cuda.current_stream().cuda_stream
dbg = os.environ.pop("STEMGNN_WG_DEBUG", "16")
os.environ["STEMGNN_WG_DEBUG"] = "0"
# the debug level is read per launch: warm up silently, then one traced launch
for _ in range(3):
    lib.stemgnn_spectral_glu_bwd(packed.data_ptr(), saved.data_ptr(), scratch.data_ptr(), gradpart.data_ptr(), nsplit, 2,
                                 B, N, W, multi, st)
torch.cuda.synchronize()
os.environ["STEMGNN_WG_DEBUG"] = dbg
lib.stemgnn_spectral_glu_bwd(packed.data_ptr(), saved.data_ptr(), scratch.data_ptr(), gradpart.data_ptr(), nsplit, 2,
                             B, N, W, multi, st)
torch.cuda.synchronize()
