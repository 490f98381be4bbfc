"""GPU parity of the fused callers around the blocks: fc tail (models/base_model.py:175-179) and the
RMSprop step of the reference driver (models/handler.py:127,165)."""
import pytest
import torch

from tests.util import relerr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


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


def test_fused_adam_matches_torch_adam():
    """models/handler.py:128-129 branch: FusedAdam (flat buffers, device-side step count and lr) vs torch.optim.Adam over
    several steps with an ExponentialLR scheduler in between, fused zero_grad semantics included."""
    from stemgnn_amd.optim import FusedAdam

    torch.manual_seed(1)
    shapes = [(7, 5), (13,), (1, 4, 1, 6, 6), (3,), (129, 31)]
    ref_p = [torch.randn(s, requires_grad=True) for s in shapes]
    my_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-3, betas=(0.9, 0.999))
    my_opt = FusedAdam(my_p, lr=1e-3, betas=(0.9, 0.999))
    sched_r = torch.optim.lr_scheduler.ExponentialLR(ref_opt, gamma=0.5)
    sched_m = torch.optim.lr_scheduler.ExponentialLR(my_opt, gamma=0.5)
    for it in range(6):
        for rp, mp in zip(ref_p, my_p):
            g = torch.randn(rp.shape)
            rp.grad = g.clone()
            mp.grad.copy_(g.cuda())                      # grads live in the flat bucket views
        ref_opt.step()
        my_opt.step()
        if it == 2:
            sched_r.step(); sched_m.step()
        for mp in my_p:
            assert float(mp.grad.abs().max()) == 0.0     # fused zero_grad
    torch.cuda.synchronize()
    for rp, mp in zip(ref_p, my_p):
        assert relerr(mp.detach(), rp.detach()) < 2e-6


def test_train_step_with_fused_adam_captures_a_graph_and_matches_eager():
    """engine.TrainStep with FusedAdam: the whole step is one hipGraph (device-side step count), and graph replays give
    the parameters an eager run of the same steps gives (capture warm-ups rolled back, incl. Adam's moments and count)."""
    from stemgnn_amd import Model
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedAdam
    N, W, H, multi, B, T = 20, 12, 3, 5, 4, 120
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    series = torch.randn(T, N, generator=g).to(dev)
    hi = (torch.randint(0, T - W - H, (6, B), generator=g) + W).to(dev)
    outs = []
    for graph in (True, False):
        torch.manual_seed(7)
        model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0).to(dev).train()
        opt = FusedAdam(model.parameters(), lr=1e-3)
        step = TrainStep(model, opt, B, W, H, N, series=series, graph=graph)
        for i in range(6):
            step.run_indices(hi[i])
        torch.cuda.synchronize()
        assert step.mode.startswith("hipgraph") == graph, step.mode
        outs.append(opt.flat_p.clone())
    assert relerr(outs[0], outs[1]) < 1e-6


def test_train_step_queue_mode_equals_per_step_indices():
    """TrainStep(order_capacity=...): load_order once + run_next per step (device-side iterator, graph replay only) gives
    the parameters run_indices gives on the same batches -- eager first step, capture warm-ups (rolled back, iterator
    included) and replays; stepping past the loaded order raises on the host."""
    from stemgnn_amd import Model, ops
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop
    N, W, H, multi, B, T = 20, 12, 3, 5, 4, 120
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    series = torch.randn(T, N, generator=g).to(dev)
    hi = (torch.randint(0, T - W - H, (6, B), generator=g) + W).to(dev)
    outs = []
    for queue in (True, False):
        torch.manual_seed(7)
        model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0).to(dev).train()
        opt = FusedRMSprop(model.parameters(), lr=1e-3)
        step = TrainStep(model, opt, B, W, H, N, series=series, graph=True, order_capacity=6 * B if queue else 0)
        if queue:
            step.load_order(hi)
            for i in range(6):
                step.run_next()
            with pytest.raises(IndexError):
                step.run_next()
        else:
            for i in range(6):
                step.run_indices(hi[i])
        torch.cuda.synchronize()
        assert step.mode.startswith("hipgraph"), step.mode
        ops.check_gather_status(dev)
        outs.append((opt.flat_p.clone(), float(step.epoch_loss_sum())))
        if queue:
            # mixing the two iterators: a full batch by explicit index while a loaded order still has batches
            # left would silently discard them -- refused; once the order is used up it is a one-batch order of its own
            step.load_order(hi[:2])
            with pytest.raises(RuntimeError, match="still queued"):
                step.run_indices(hi[0])
            step.run_next()
            step.run_next()
            keep = opt.flat_p.clone()
            step.run_indices(hi[0])
            torch.cuda.synchronize()
            assert not torch.equal(keep, opt.flat_p)
            ops.check_gather_status(dev)
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


@pytest.mark.parametrize("B,N,W,H", [(32, 228, 12, 3), (5, 33, 12, 1), (3, 50, 8, 4), (16, 64, 48, 12)])
def test_fused_train_tail_matches_separate_stages(B, N, W, H):
    """stemgnn_fc_tail_train (fc fwd + MSE + both backwards, 2 launches) == FcTail -> MSELoss -> backward (5 launches):
    loss, d(loss)/d(fsum) and the four fc gradients, also with a non-unit upstream gradient."""
    from stemgnn_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N)
    fsum = torch.randn(B, N, W, generator=g).to(DEV)
    y = torch.randn(B, H, N, generator=g).to(DEV)
    prm = [torch.randn(W, W, generator=g).to(DEV) * 0.3, torch.randn(W, generator=g).to(DEV) * 0.1,
           torch.randn(H, W, generator=g).to(DEV) * 0.3, torch.randn(H, generator=g).to(DEV) * 0.1]

    def run(fused, scale):
        f = fsum.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in prm]
        if fused:
            loss = ops.FcTailMse.apply(f, y, *ps)
        else:
            loss = ops.mse_loss(ops.FcTail.apply(f, *ps), y)
        (scale * loss).backward()
        return loss.detach(), f.grad, [p.grad for p in ps]

    for scale in (1.0, 2.5):
        l0, df0, g0 = run(False, scale)
        l1, df1, g1 = run(True, scale)
        assert abs(float(l0) - float(l1)) <= 2e-6 * abs(float(l0))
        assert float((df0 - df1).abs().max()) <= 2e-6 * float(df0.abs().max())
        for a, b in zip(g0, g1):
            assert float((a - b).abs().max()) <= 5e-6 * max(float(a.abs().max()), 1e-6)
    acc = torch.zeros((), device=DEV, dtype=torch.float64)
    out = torch.zeros((), device=DEV)
    l2 = ops.FcTailMse.apply(fsum, y, *prm, None, out, acc)
    assert float(out) == float(l2) and abs(float(acc) - float(l2)) < 1e-12


@pytest.mark.parametrize("B,N,W,H", [(32, 228, 12, 3), (5, 33, 12, 1), (3, 50, 28, 28)])
def test_fc_tail_train_in_two_calls_equals_the_one_call(B, N, W, H):
    """Round 6: stemgnn_fc_tail_train_rows + _finish (the step queues `_finish` on the side branch: nothing on the backward's
    chain reads the loss or the fc gradients) are the one call's two launches: bit-identical loss, loss accumulator, d(fsum) and
    fc gradients, whichever stream `_finish` runs on."""
    from stemgnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 3 + H)
    fsum, target = torch.randn(B, N, W, generator=g).cuda(), torch.randn(B, H, N, generator=g).cuda()
    w0, b0 = (torch.randn(W, W, generator=g) * 0.3).cuda(), (torch.randn(W, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn(H, W, generator=g) * 0.3).cuda(), (torch.randn(H, generator=g) * 0.1).cuda()
    st = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()

    def buffers():
        return dict(scratch=torch.empty(lib.stemgnn_fc_tail_train_scratch_floats(B, N, W, H), device=DEV),
                    loss=torch.zeros((), device=DEV), acc=torch.full((), 0.5, device=DEV, dtype=torch.float64),
                    dfsum=torch.empty_like(fsum), dw0=torch.empty_like(w0), db0=torch.empty_like(b0),
                    dw2=torch.empty_like(w2), db2=torch.empty_like(b2))
    one, two = buffers(), buffers()
    _lib.check(lib.stemgnn_fc_tail_train(fsum.data_ptr(), target.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(),
                                         b2.data_ptr(), B, N, W, H, one["scratch"].data_ptr(), None, one["loss"].data_ptr(),
                                         one["acc"].data_ptr(), one["dfsum"].data_ptr(), one["dw0"].data_ptr(),
                                         one["db0"].data_ptr(), one["dw2"].data_ptr(), one["db2"].data_ptr(), st), "one call")
    _lib.check(lib.stemgnn_fc_tail_train_rows(fsum.data_ptr(), target.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(),
                                              b2.data_ptr(), B, N, W, H, two["scratch"].data_ptr(), None, two["dfsum"].data_ptr(),
                                              st), "rows")
    side.wait_stream(torch.cuda.current_stream())
    _lib.check(lib.stemgnn_fc_tail_train_finish(two["scratch"].data_ptr(), B, N, W, H, two["loss"].data_ptr(), two["acc"].data_ptr(),
                                                two["dw0"].data_ptr(), two["db0"].data_ptr(), two["dw2"].data_ptr(),
                                                two["db2"].data_ptr(), side.cuda_stream), "finish")
    torch.cuda.synchronize()
    for k in ("loss", "acc", "dfsum", "dw0", "db0", "dw2", "db2"):
        assert torch.equal(one[k], two[k]), k
    assert float(one["loss"]) > 0 and abs(float(one["acc"]) - 0.5 - float(one["loss"])) < 1e-6
    assert lib.stemgnn_fc_tail_train_finish(None, B, N, W, H, two["loss"].data_ptr(), None, two["dw0"].data_ptr(),
                                            two["db0"].data_ptr(), two["dw2"].data_ptr(), two["db2"].data_ptr(), st) == _lib.SG_EINVAL


@pytest.mark.parametrize("offset,nbytes", [(0, 4096), (4, 4), (4, 60), (12, 4096 + 8), (0, 16), (8, 1 << 20), (0, 0)])
def test_fill_zero_is_a_kernel_with_exact_edges(offset, nbytes):
    """Round 6: the step path zeroes through stemgnn_fill_zero / sg_zero_async (a fill KERNEL: this HIP runtime mis-replays
    memset nodes of a captured graph).  Ragged heads / tails: exactly [offset, offset + nbytes) is cleared,
    the neighbours keep their bytes -- eagerly and replayed from a hipGraph five times."""
    from stemgnn_amd import _lib
    lib = _lib.load()
    buf = torch.full(((1 << 20) + 4096,), 7.0, device=DEV)
    st = torch.cuda.current_stream()

    def check():
        torch.cuda.synchronize()
        lo, hi = offset // 4, (offset + nbytes) // 4
        assert bool((buf[:lo] == 7.0).all()) and bool((buf[hi:] == 7.0).all())
        assert bool((buf[lo:hi] == 0.0).all())
    _lib.check(lib.stemgnn_fill_zero(buf.data_ptr() + offset, nbytes, st.cuda_stream), "fill_zero")
    check()
    s = torch.cuda.Stream()
    s.wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            buf.fill_(7.0)                                                   # a kernel node, then the fill kernel
            _lib.check(lib.stemgnn_fill_zero(buf.data_ptr() + offset, nbytes, torch.cuda.current_stream().cuda_stream), "fill_zero")
    for _ in range(5):
        g.replay()
        check()
