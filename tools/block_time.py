"""Per-stage timing of one StockBlock (forward + backward C-ABI stages) at a given shape, HIP events on the launch
stream.  GPU tool:  python tools/block_time.py [B N W multi] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stemgnn_amd import _lib, ops  # noqa: E402
from stemgnn_amd.base_model import StockBlockLayer  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, N, W, multi = (a + [32, 228, 12, 5][len(a):])[:4] if len(a) < 4 else a[:4]
    iters = a[4] if len(a) > 4 else 30
    nsplit = ops._NSPLIT
    lib = _lib.load()
    dev, f32 = torch.device("cuda"), torch.float32
    torch.manual_seed(0)
    blk = StockBlockLayer(W, N, multi, stack_cnt=0).to(dev)
    params = blk.hip_params()
    parr = _lib.ptr_array(params)
    X = torch.randn(B, N, W, device=dev)
    mul_L = torch.randn(4, N, N, device=dev) * 0.05
    mul_L[0].zero_()
    tables = ops.dft_tables(W, multi, dev)
    pk = torch.empty(lib.stemgnn_packed_floats(W, multi), device=dev, dtype=f32)
    sv = torch.empty(lib.stemgnn_saved_floats(B, N, W, multi), device=dev, dtype=f32)
    forecast = torch.empty(B, N, W, device=dev)
    backcast = torch.empty(B, N, W, device=dev)
    scratch = torch.empty(lib.stemgnn_scratch_floats(B, N, W, multi), device=dev, dtype=f32)
    dG = scratch[lib.stemgnn_scratch_offset_dG(B, N, W, multi):]
    gradpart = torch.empty(lib.stemgnn_gradpart_floats(W, multi, nsplit), device=dev, dtype=f32)
    dmul_L = torch.zeros(4, N, N, device=dev)
    dX = torch.empty(B, N, W, device=dev)
    dforecast = torch.randn(B, N, W, device=dev)
    dbackcast = torch.randn(B, N, W, device=dev)
    grads = [None if p is None else torch.empty_like(p) for p in params]
    garr = _lib.ptr_array(grads)
    st = torch.cuda.current_stream()
    s = st.cuda_stream
    sb, sn, stt = N * W, W, 1
    P = lambda t: t.data_ptr()  # noqa: E731
    stages = [
        ("block_pack", lambda: lib.stemgnn_block_pack(parr, P(tables), P(pk), W, multi, s)),
        ("gft_fwd", lambda: lib.stemgnn_gft_fwd(P(mul_L), P(X), sb, sn, stt, P(sv), B, N, W, s)),
        ("spectral_glu_fwd", lambda: lib.stemgnn_spectral_glu_fwd(P(pk), P(sv), B, N, W, multi, s)),
        ("igft_heads_fwd", lambda: lib.stemgnn_igft_heads_fwd(parr, P(pk), P(sv), P(X), sb, sn, stt, P(forecast), 0,
                                                               P(backcast), B, N, W, multi, s)),
        ("igft_heads_bwd data", lambda: lib.stemgnn_igft_heads_bwd(
            parr, P(pk), P(sv), P(X), sb, sn, stt, P(dforecast), P(dbackcast), P(backcast), P(scratch), P(gradpart),
            nsplit, 1, B, N, W, multi, s)),
        ("igft_heads_bwd wgrad", lambda: lib.stemgnn_igft_heads_bwd(
            parr, P(pk), P(sv), P(X), sb, sn, stt, P(dforecast), P(dbackcast), P(backcast), P(scratch), P(gradpart),
            nsplit, 2, B, N, W, multi, s)),
        ("spectral_glu_bwd data", lambda: lib.stemgnn_spectral_glu_bwd(P(pk), P(sv), P(scratch), P(gradpart), nsplit, 1,
                                                                       B, N, W, multi, s)),
        ("spectral_glu_bwd wgrad", lambda: lib.stemgnn_spectral_glu_bwd(P(pk), P(sv), P(scratch), P(gradpart), nsplit, 2,
                                                                        B, N, W, multi, s)),
        ("gft_bwd", lambda: lib.stemgnn_gft_bwd(P(mul_L), P(X), sb, sn, stt, P(dG), P(dX), P(dmul_L), 0, B, N, W, s)),
        ("block_wgrad (all weight gradients)", lambda: lib.stemgnn_block_wgrad(
            parr, P(pk), P(sv), P(X), sb, sn, stt, P(dforecast), 1, P(scratch), P(gradpart), nsplit, 100, B, N, W, multi, s)),
        ("block_unpack_grads", lambda: lib.stemgnn_block_unpack_grads(P(gradpart), nsplit, P(tables), garr, W, multi, 1, s)),
    ]
    total = 0.0
    for name, fn in stages:
        for _ in range(3):
            _lib.check(fn(), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        total += us
        print(f"{name:26s} {us:9.1f} us")
    print(f"{'sum (one block fwd+bwd)':26s} {total:9.1f} us   B={B} N={N} W={W} multi={multi} nsplit={nsplit}")


if __name__ == "__main__":
    main()
