"""Data-parallel gradient synchronisation for the StemGNN hot path (new work: the reference has no
distributed code at all).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm).  The path shards over
the batch with no data-path collective ("replicas with a local graph"): every rank runs the
reference math on its own batch shard; the only exchange is ONE flat fp32 all-reduce of the gradients
per optimizer step (4.9 MB at PEMS07 -- latency-bound on xGMI).  Round 4: inside the captured step the flat buffer is
reduced as TWO contiguous ranges -- the spectral blocks' + fc gradients (4.2 MB, complete when block 1's un-packing has run)
on the side branch under the GRU backward recurrence, the GRU / attention gradients (0.67 MB) behind the GRU weight
gradients -- so only the small one is exposed (engine.TrainStep).
"""
import torch
import torch.distributed as dist


class FlatGradBucket:
    """All parameter gradients as views into one contiguous fp32 buffer -> a single all-reduce per step.

    Parameters that never receive a gradient (stock_block.1.backcast_short_cut.*, reference
    models/base_model.py:73-74) keep an all-zero slot, so the bucket layout is identical on every rank.
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        # optional stand-in for the collective: reduce_fn(view) runs IN-STREAM on the current stream in place of
        # dist.all_reduce(view, SUM) -- the hook the schedule tests use to put a collective that CHANGES data on a 1-GPU box
        # (tests/test_hip_schedule.py); reduce_fn.world, if present, is the world size it stands for
        self.reduce_fn = None
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.attach()

    def attach(self):
        """(Re)bind p.grad to the flat views; backward then accumulates in place (graph-capturable)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        if self.flat.is_cuda:
            # a kernel node, not tensor.zero_() (= a memset node once the step is captured into a hipGraph: csrc/devattr.h)
            from . import _lib
            _lib.check(_lib.load().stemgnn_fill_zero(self.flat.data_ptr(), self.flat.numel() * self.flat.element_size(),
                                                     torch.cuda.current_stream(self.flat.device).cuda_stream), "fill_zero")
        else:
            self.flat.zero_()

    def all_reduce_sum(self, group=None, force=False):
        """SUM over the ranks; the consumer scales by 1 / world (FusedRMSprop.grad_scale: inside the optimizer kernel,
        no separate averaging pass over the bucket).  `force`: issue the collective even in a one-rank group (RCCL
        readiness tests on a 1-GPU box: same call, same stream semantics, result unchanged)."""
        if self.flat.is_cuda:
            from .ops import join_side_streams
            join_side_streams(self.flat.device)      # weight gradients may still be in flight on the side stream
        return self._reduce(self.flat, group, force)

    def _reduce(self, view, group, force):
        if self.reduce_fn is not None:
            self.reduce_fn(view)
            return int(getattr(self.reduce_fn, "world", 1))
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
            return dist.get_world_size(group)
        return 1

    def offset_of(self, param):
        """Offset (in elements) of `param`'s gradient view inside the flat buffer."""
        off = 0
        for p in self.params:
            if p is param:
                return off
            off += p.numel()
        raise ValueError("parameter is not in this bucket")

    def all_reduce_range(self, lo, hi, group=None, force=False):
        """SUM of flat[lo:hi] over the ranks on the CURRENT stream (no side-stream join: the caller orders it behind the
        kernels that produce that range -- ops.SpectralHotPath.backward calls it on the side stream right behind block 1's
        un-packing).  Same one-rank `force` semantics as all_reduce_sum."""
        if hi <= lo:
            return 1
        return self._reduce(self.flat[lo:hi], group, force)

    def all_reduce_mean(self, group=None):
        world = self.all_reduce_sum(group)
        if world > 1:
            self.flat.div_(world)


def shard_batch(global_batch, rank, world_size):
    """Contiguous [start, stop) shard of a global batch (ragged tails go to the low ranks)."""
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank `src`'s weights."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
