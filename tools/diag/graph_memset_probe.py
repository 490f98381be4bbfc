"""Is a memset node of a captured hipGraph ordered ahead of the kernel node that follows it on the same stream?

Round 6 found the serialised capture of the train step wrong because the FIRST hipMemsetAsync of the graph (arrival counters of a
weight-gradient launch) took effect while its consumer kernel was already running.  This probe asks the same of plain torch
ops, no stemgnn kernels involved: on ONE captured stream

    [warm kernels] -> b.fill_(5) (kernel) -> b.zero_() (memset node) -> c = b + 1 (kernel) -> ... repeated `pairs` times

After a replay every c must be 1; a c of 6 means the kernel read b before the memset node's fill landed.
usage: graph_memset_probe.py [n_replays]
"""
import ctypes
import sys

import torch

_hip = ctypes.CDLL("libamdhip64.so")
_hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]


def memset0(t):
    """a REAL memset node (tensor.zero_() is a fill kernel on a HIP device)"""
    rc = _hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def case(pairs, numel, warm, replays, real=True):
    dev = torch.device("cuda:0")
    a = torch.randn(2048, 2048, device=dev)
    bs = [torch.empty(numel, device=dev) for _ in range(pairs)]
    cs = [torch.empty(numel, device=dev) for _ in range(pairs)]

    def body():
        x = a
        for _ in range(warm):
            x = x @ a * 1e-3
        for b, c in zip(bs, cs):
            b.fill_(5.0)
            memset0(b) if real else b.zero_()
            torch.add(b, 1.0, out=c)
        return x

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    bad = [0] * pairs
    seen = set()
    for _ in range(replays):
        for c in cs:
            c.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        for i, c in enumerate(cs):
            if not bool((c == 1.0).all()):
                bad[i] += 1
                seen.add(float(c[0]))
    return bad, sorted(seen)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    print("torch", torch.__version__, "hip", torch.version.hip)
    for pairs, numel, warm in ((1, 64, 0), (1, 64, 4), (3, 64, 4), (3, 1 << 17, 4), (3, 1 << 22, 4), (6, 100, 1)):
        bad, seen = case(pairs, numel, warm, n)
        print(f"pairs={pairs} numel={numel} warm={warm}: replays with a wrong c, per pair: {bad} of {n}; wrong values seen {seen}")
    import os
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
