"""split-bf16 vs exact-fp32 MFMA on the GLU layer shapes of the PEMS07 configuration (both branches of a
layer are two such products).  Prints time per product (HIP events) and the norm-relative error against fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from stemgnn_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, what) in ((7296, 480, 240, "GLU layer 1/2, PEMS07 (M = 32 x 228)"), (7296, 480, 36, "GLU layer 0 (K = 3W after the DFT fold)"),
                        (11456, 480, 240, "PEMS03 (M = 32 x 358)"), (32768, 1920, 960, "configs[4] shard (M = 16 x 2048, W = 48)")):
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    ref = (A.double() @ B.double().T)
    den = float(ref.abs().max())
    C = torch.empty(M, N, device=dev)
    t32 = timed(lambda: lib.stemgnn_glu_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st))
    e32 = float((C.double() - ref).abs().max()) / den
    flops = 2.0 * M * N * K
    line = f"{what}: M={M} N={N} K={K}\n  exact fp32 MFMA : {t32:8.1f} us  {flops / t32 / 1e6:7.1f} TFLOP/s  err {e32:.1e}"
    for s in (3, 2, 1):
        planes = torch.empty(lib.stemgnn_split_planes_floats(N, K, s), device=dev)
        lib.stemgnn_split_weights_bf16(B.data_ptr(), N, K, s, planes.data_ptr(), st)
        t = timed(lambda: lib.stemgnn_glu_gemm_bf16(A.data_ptr(), planes.data_ptr(), C.data_ptr(), M, N, K, s, st))
        err = float((C.double() - ref).abs().max()) / den
        line += f"\n  bf16 x {s} ({[0, 1, 3, 6][s]} products): {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s-equivalent  err {err:.1e}  speed-up {t32 / t:.2f}x"
    print(line, flush=True)
