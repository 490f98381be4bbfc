"""Split-bf16 GLU GEMM (csrc/splitgemm.hip) against an fp64 product: splits = 3 must be fp32-class (the parity budget of
the hot path with a wide margin), 2 inside 1e-4 norm-relative, 1 is plain bf16; and the exact-fp32 core at the same shape."""
import pytest
import torch

from tests.util import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(7296, 480, 240), (300, 100, 36), (64, 128, 32), (1000, 256, 244)])
def test_split_bf16_gemm_accuracy(M, N, K):
    from stemgnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    ref = A.double() @ B.double().T
    st = torch.cuda.current_stream().cuda_stream
    C = torch.empty(M, N, device=dev)
    _lib.check(lib.stemgnn_glu_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st), "f32")
    e32 = relerr(C, ref)
    errs = {}
    for s in (3, 2, 1):
        planes = torch.empty(lib.stemgnn_split_planes_floats(N, K, s), device=dev)
        _lib.check(lib.stemgnn_split_weights_bf16(B.data_ptr(), N, K, s, planes.data_ptr(), st), "split")
        C.zero_()
        _lib.check(lib.stemgnn_glu_gemm_bf16(A.data_ptr(), planes.data_ptr(), C.data_ptr(), M, N, K, s, st), "bf16")
        errs[s] = relerr(C, ref)
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: fp32 MFMA {e32:.2e} | bf16x3 {errs[3]:.2e} | bf16x2 {errs[2]:.2e} | bf16 {errs[1]:.2e}")
    assert e32 < 2e-6
    assert errs[3] < 4e-6          # fp32 class
    assert errs[2] < 1e-4          # inside the parity budget, ~1e-5 expected
    assert errs[1] < 3e-2
