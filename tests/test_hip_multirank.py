"""The N > 1 training step on real HIP kernels: two ranks sharing the one GPU of the test box, gloo as the transport
(RCCL refuses two ranks on one device; the all-reduce sits between the two hipGraphs either way).  Exercises what the
multi-GPU bench runs: parameter broadcast, engine.TrainStep(world=2) = hipGraph(gather+fwd+bwd) -> flat gradient
all-reduce -> hipGraph(optimizer), replicas staying bit-identical, and the averaged gradient being what is applied."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
CFG = dict(N=20, W=12, H=3, multi=5, B=4, T=200)     # hidden size 20: single-workgroup GRU clusters (no co-residency needs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stemgnn_amd import Model
    from stemgnn_amd.distributed import broadcast_parameters
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop
    c = CFG
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)                               # different initial weights per rank on purpose
    model = Model(c["N"], 2, c["W"], c["multi"], horizon=c["H"], dropout_rate=0.0).to(dev).train()
    broadcast_parameters(model)
    opt = FusedRMSprop(model.parameters(), lr=1e-3, eps=1e-8)
    g = torch.Generator().manual_seed(5 + rank)                 # every rank trains on its own windows
    series = torch.randn(c["T"], c["N"], generator=g).to(dev)
    step = TrainStep(model, opt, c["B"], c["W"], c["H"], c["N"], series=series, world=world, graph=True)
    hi = (torch.randint(0, c["T"] - c["W"] - c["H"], (8, c["B"]), generator=g) + c["W"]).to(dev)
    # step 0 by hand: the gradient every rank applies must be the mean of the two local gradients
    p0 = opt.flat_p.clone()
    x, y = series.new_empty(c["B"], c["W"], c["N"]), series.new_empty(c["B"], c["H"], c["N"])
    from stemgnn_amd import ops
    ops.window_gather(series, hi[0], c["W"], c["H"], x, y)
    forecast, _ = model(x)
    ops.mse_loss(forecast, y).backward()
    ops.join_side_streams(dev)
    local = opt.bucket.flat.clone()
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    mean_grad = (both[0] + both[1]) / world
    opt.bucket.zero()
    losses = []
    for i in range(8):                                          # eager first step, capture, replays
        step.run_indices(hi[i])
        losses.append(float(step.loss))
        if i == 0:
            sq = 0.01 * mean_grad * mean_grad                   # RMSprop from a zero state
            want = p0 - 1e-3 * mean_grad / (sq.sqrt() + 1e-8)
            err = float((opt.flat_p - want).abs().max() / want.abs().max())
            out[("step0", rank)] = err
    torch.cuda.synchronize()
    out[("mode", rank)] = step.mode
    out[("params", rank)] = opt.flat_p.cpu()
    out[("losses", rank)] = losses
    dist.destroy_process_group()


def test_two_rank_train_step_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        assert out[("mode", r)].startswith("hipgraph(fwd+bwd)"), out[("mode", r)]
        assert out[("step0", r)] < 1e-6, out[("step0", r)]     # the averaged gradient is what the optimizer applied
        assert all(l == l and l < 1e3 for l in out[("losses", r)])
    assert torch.equal(out[("params", 0)], out[("params", 1)])  # replicas stay bit-identical through graph replays


def _exact_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stemgnn_amd import Model, ops
    from stemgnn_amd.distributed import FlatGradBucket, broadcast_parameters, shard_batch
    N, W, H, multi, B = 24, 12, 3, 5, 8
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(3)
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0).to(dev).train()
    broadcast_parameters(model)
    bucket = FlatGradBucket(model.parameters())
    ops.set_direct_grad(model, True, overlap=True)
    model.hot_state.exact_group = (None, world)                 # A / dA averaged over the ranks
    torch.manual_seed(4)
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)          # the same GLOBAL batch on every rank
    lo, hi = shard_batch(B, rank, world)
    forecast, att = model(x[lo:hi].to(dev))
    ops.mse_loss(forecast, y[lo:hi].to(dev)).backward()
    scale = 1.0 / bucket.all_reduce_sum()                       # what FusedRMSprop.grad_scale applies in the kernel
    torch.cuda.synchronize()
    out[("grad", rank)] = (bucket.flat * scale).cpu()
    out[("att", rank)] = att.detach().cpu()
    if rank == 0:
        out["sd"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    dist.destroy_process_group()


def test_exact_mode_split_batch_equals_single_process():
    """Exact data-parallel mode on the real kernels (two ranks on the one GPU, gloo transport): with the [N,N] attention
    mean all-reduced in forward and its gradient in backward, a global batch of 8 split 4 + 4 yields the attention
    matrix and the (rank-averaged) gradient of EVERY parameter that one process computes on all 8 samples."""
    from stemgnn_amd import Model, ops
    from stemgnn_amd.distributed import FlatGradBucket
    from tests.util import relerr
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exact_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.equal(out[("grad", 0)], out[("grad", 1)])
    N, W, H, multi, B = 24, 12, 3, 5, 8
    dev = torch.device("cuda:0")
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0).to(dev).train()
    model.load_state_dict(out["sd"])
    bucket = FlatGradBucket(model.parameters())
    ops.set_direct_grad(model, True, overlap=False)
    torch.manual_seed(4)
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)
    forecast, att = model(x.to(dev))
    ops.mse_loss(forecast, y.to(dev)).backward()
    torch.cuda.synchronize()
    assert relerr(out[("att", 0)], att) < 1e-5 and relerr(out[("att", 1)], att) < 1e-5
    off = 0
    for (k, p), v in zip(model.named_parameters(), bucket.views):
        got = out[("grad", 0)][off:off + p.numel()].view_as(p)
        off += p.numel()
        if float(v.abs().max()) == 0.0:
            assert float(got.abs().max()) == 0.0, k
        else:
            assert relerr(got, v) < 2e-5, (k, relerr(got, v))
