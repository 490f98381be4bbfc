"""Fused optimizer step for the reference driver's training loop.

The reference trains with ``torch.optim.RMSprop(params, lr, eps=1e-8)`` (models/handler.py:127) and calls
``model.zero_grad()`` / ``optim.step()`` every batch (:160,:165): ~70 small tensors, i.e. a launch-bound cloud of
tiny kernels.  ``FusedRMSprop`` keeps ALL parameters, gradients and the running square average in three flat
buffers and performs the step -- and the zeroing of the gradients for the next step -- in ONE HIP kernel
(``stemgnn_rmsprop_step``).  Same arithmetic as torch.optim.RMSprop(momentum=0, centered=False, weight_decay=0).
"""
import torch

from . import _lib
from .distributed import FlatGradBucket


class FusedRMSprop(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, bucket=None, fuse_zero_grad=True):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps))
        if not params or not params[0].is_cuda:
            raise _lib.StemGNNHipError("FusedRMSprop needs parameters on a HIP device (move the model first)")
        dev = params[0].device
        self._params = params
        self.numel = sum(p.numel() for p in params)
        self.flat_p = torch.empty(self.numel, device=dev, dtype=torch.float32)
        off = 0
        for p in params:                      # re-point every parameter at its slice of the flat buffer
            n = p.numel()
            view = self.flat_p[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += n
        self.bucket = bucket if bucket is not None else FlatGradBucket(params)
        if self.bucket.numel != self.numel:
            raise ValueError("gradient bucket and parameter list differ")
        self.square_avg = torch.zeros_like(self.flat_p)
        self._lr_host = float(lr)
        self._lr_dev = torch.tensor([lr], device=dev, dtype=torch.float32)
        self.fuse_zero_grad = bool(fuse_zero_grad)
        self.grad_scale = 1.0      # applied to every gradient inside the kernel (1 / world after a SUM all-reduce)

    def zero_grad(self, set_to_none=False):   # gradients live in the flat bucket; never drop the views
        self.bucket.zero()

    def _check_grad_views(self):
        """The step reads the FLAT gradient buffer: every p.grad must still be its view (torch's default
        model.zero_grad(set_to_none=True) drops them -> backward would fill fresh tensors and the step would see zeros).
        Object identity first (what attach() stored; a handful of ns per parameter on the launch-bound eager path), the
        device-pointer comparison only for a .grad someone replaced."""
        for p, view in zip(self.bucket.params, self.bucket.views):
            g = p.grad
            if g is view:
                continue
            if g is None or g.data_ptr() != view.data_ptr():
                raise _lib.StemGNNHipError(
                    f"{type(self).__name__}: a parameter's .grad is no longer the flat-bucket view (model.zero_grad() with "
                    "set_to_none=True?).  Use optimizer.zero_grad() / model.zero_grad(set_to_none=False), or call "
                    "optimizer.bucket.attach() before backward.")

    def sync_lr(self):
        """Push a learning rate an LR scheduler changed (host side) to the device word the kernel reads.  step() calls
        it; a hipGraph replay does not run step()'s Python, so engine.TrainStep calls it before every replay."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self._lr_host = lr
            self._lr_dev.fill_(lr)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        self.sync_lr()
        from .ops import join_side_streams
        join_side_streams(self.flat_p.device)
        self._check_grad_views()
        lib = _lib.load()
        _lib.check(lib.stemgnn_rmsprop_step(
            self.flat_p.data_ptr(), self.bucket.flat.data_ptr(), self.square_avg.data_ptr(), self.numel,
            self._lr_dev.data_ptr(), float(group["alpha"]), float(group["eps"]), int(self.fuse_zero_grad),
            float(self.grad_scale), torch.cuda.current_stream().cuda_stream), "rmsprop_step")
        return loss


class FusedAdam(FusedRMSprop):
    """``torch.optim.Adam(params, lr, betas=(0.9, 0.999))`` of the driver's other optimizer branch (models/handler.py:128-129)
    over the same flat buffers as FusedRMSprop: one kernel (+ a one-thread step-count tick), gradient zeroing fused in,
    lr and the step count in device memory -- so an Adam run keeps the hipGraph train step too."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, bucket=None, fuse_zero_grad=True):
        params = [p for p in params if p.requires_grad]
        FusedRMSprop.__init__(self, params, lr=lr, alpha=0.0, eps=eps, bucket=bucket, fuse_zero_grad=fuse_zero_grad)
        self.param_groups[0]["betas"] = tuple(betas)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = self.square_avg                       # reuse the second flat state buffer
        self._step_dev = torch.zeros(1, device=self.flat_p.device, dtype=torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        group = self.param_groups[0]
        self.sync_lr()
        from .ops import join_side_streams
        join_side_streams(self.flat_p.device)
        self._check_grad_views()
        b1, b2 = group["betas"]
        lib = _lib.load()
        _lib.check(lib.stemgnn_adam_step(
            self.flat_p.data_ptr(), self.bucket.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            self.numel, self._lr_dev.data_ptr(), self._step_dev.data_ptr(), float(b1), float(b2), float(group["eps"]),
            int(self.fuse_zero_grad), float(self.grad_scale), torch.cuda.current_stream().cuda_stream), "adam_step")
        return loss
