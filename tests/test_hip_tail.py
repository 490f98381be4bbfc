"""GPU parity of the fused callers around the blocks: fc tail (models/base_model.py:175-179) and the
RMSprop step of the reference driver (models/handler.py:127,165)."""
import pytest
import torch

from tests.util import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,W,H", [(32, 228, 12, 3), (5, 33, 12, 1), (3, 50, 28, 28), (2, 7, 5, 2), (300, 3, 16, 4)])
def test_fc_tail_fwd_bwd_vs_torch(B, N, W, H):
    from stemgnn_amd.ops import FcTail

    torch.manual_seed(B + N)
    fc = torch.nn.Sequential(torch.nn.Linear(W, W), torch.nn.LeakyReLU(), torch.nn.Linear(W, H))
    fsum = torch.randn(B, N, W, requires_grad=True)
    dy = torch.randn(B, H, N)
    ref = fc(fsum).permute(0, 2, 1).contiguous()                 # reference :175-179 ([B,1,N] when H == 1)
    ref.backward(dy)
    f2 = fsum.detach().clone().cuda().requires_grad_(True)
    ps = [p.detach().clone().cuda().requires_grad_(True) for p in (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias)]
    out = FcTail.apply(f2, *ps)
    out.backward(dy.cuda())
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert relerr(out, ref.detach()) < 1e-5
    assert relerr(f2.grad, fsum.grad) < 1e-5
    for mine, theirs in zip(ps, (fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias)):
        assert relerr(mine.grad, theirs.grad) < 1e-5


def test_fused_rmsprop_matches_torch_rmsprop():
    from stemgnn_amd.optim import FusedRMSprop

    torch.manual_seed(0)
    shapes = [(7, 5), (13,), (1, 4, 1, 6, 6), (3,), (129, 31)]
    ref_p = [torch.randn(s, requires_grad=True) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref_opt = torch.optim.RMSprop(ref_p, lr=1e-3, eps=1e-8)      # handler.py:127 form (torch defaults otherwise)
    my_opt = FusedRMSprop(my_p, lr=1e-3, alpha=0.99, eps=1e-8)
    sched = torch.optim.lr_scheduler.ExponentialLR(my_opt, gamma=0.5)     # handler.py:130
    ref_sched = torch.optim.lr_scheduler.ExponentialLR(ref_opt, gamma=0.5)
    for it in range(6):
        grads = [torch.randn(s) for s in shapes]
        for p, g in zip(ref_p, grads):
            p.grad = g.clone()
        for view, g in zip(my_opt.bucket.views, grads):
            view.copy_(g.cuda())
        # a parameter that NEVER receives a gradient (block 1's backcast_short_cut): torch skips it, the fused kernel
        # sees an all-zero slot -> square_avg stays 0 and the parameter is unchanged: identical outcome
        ref_p[3].grad = None
        my_opt.bucket.views[3].zero_()
        ref_opt.step()
        my_opt.step()
        if it == 2:
            sched.step(); ref_sched.step()
        assert float(my_opt.bucket.flat.abs().max()) == 0.0      # gradients cleared by the fused kernel
    for mine, theirs in zip(my_p, ref_p):
        assert mine.data_ptr() >= my_opt.flat_p.data_ptr()       # parameters live in the flat buffer
        assert relerr(mine, theirs.detach()) < 1e-6
