"""CPU, world_size 2, gloo: the N>1 path (flat gradient bucket all-reduce, batch sharding, parameter
broadcast).  The data path itself has no collective (replicas with a local graph)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stemgnn_amd.distributed import FlatGradBucket, broadcast_parameters, shard_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    broadcast_parameters(net, src=0)
    for p in net[2].parameters():                       # a parameter that never gets a gradient
        pass
    bucket = FlatGradBucket(net.parameters())
    torch.manual_seed(7)
    xg, yg = torch.randn(10, 6), torch.randn(10, 3)     # the same global batch on every rank
    lo, hi = shard_batch(10, rank, world)
    bucket.zero()
    loss = torch.nn.functional.mse_loss(net[1](net[0](xg[lo:hi])), yg[lo:hi], reduction="sum") / 10
    loss.backward()                                     # net[2] unused -> its slot stays zero
    assert net[0].weight.grad.data_ptr() == bucket.views[0].data_ptr()   # grads ARE the flat views
    bucket.all_reduce_mean()
    bucket.flat.mul_(world)                             # mean -> sum of the per-shard partial losses
    out[rank] = (bucket.flat.clone(), [p.detach().clone() for p in net.parameters()])
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    f0, p0 = out[0]
    f1, p1 = out[1]
    assert torch.equal(f0, f1)                                          # every rank holds the same reduced grads
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))               # broadcast made the replicas identical
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), p0):
            p.copy_(v)
    torch.manual_seed(7)
    xg, yg = torch.randn(10, 6), torch.randn(10, 3)
    loss = torch.nn.functional.mse_loss(net[1](net[0](xg)), yg, reduction="sum") / 10
    loss.backward()
    ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in net.parameters()])
    assert torch.allclose(f0, ref, atol=1e-6)


@pytest.mark.parametrize("B,world", [(32, 8), (7, 2), (5, 4), (3, 4)])
def test_shard_batch_partitions(B, world):
    spans = [shard_batch(B, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == B
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
