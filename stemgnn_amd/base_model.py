"""Drop-in for the reference's ``models.base_model`` (microsoft/StemGNN models/base_model.py).

Same public surface -- ``Model(units, stack_cnt, time_step, multi_layer, horizon=1, dropout_rate=0.5,
leaky_rate=0.2, device='cpu')``, ``forward(x[B,W,N]) -> (forecast[B,H,N], attention[N,N])``,
``StockBlockLayer(time_step, unit, multi_layer, stack_cnt)``, ``GLU(in, out)`` -- the same
``state_dict`` keys / shapes and the same parameter creation order (so a shared ``torch.manual_seed``
gives the same initial weights, and checkpoints interchange), but the modules are only parameter
containers: all arithmetic after the GRU runs in the hand-written HIP kernels of
``libstemgnn_hip.so`` through :class:`stemgnn_amd.ops.SpectralHotPath`.

The ``nn.GRU`` module (models/base_model.py:92,137) is kept as the parameter container; its recurrence
runs in the persistent HIP kernels of ``csrc/gru.hip`` (``STEMGNN_GRU=miopen`` selects the library GRU for
A/B runs).  The 2-layer ``fc`` tail (:175-179) runs in the fused kernel of ``csrc/tail.hip``.
"""
import itertools
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import FcTail, FcTailMse, GruFront, SpectralHotPath, StockBlockFn

_instance_counter = itertools.count()
_BLOCK_FIELDS = ("forecast", "forecast_result", "backcast", "backcast_short_cut")
_FC_TAIL_MAX_W, _FC_TAIL_MAX_H = 64, 32      # stemgnn_fc_tail_supported (csrc/tail.hip): both weight matrices of a row block in LDS


class GLU(nn.Module):
    """One gated linear unit (reference :6-13).  Inside StockBlockLayer / Model the arithmetic is fused into the spectral
    GEMM epilogues and this module only holds the parameters; called on its own, ``forward`` runs the same math through
    the stand-alone composition ``ops.GluFn`` (general fp32 MFMA GEMM + elementwise HIP kernels)."""

    def __init__(self, input_channel, output_channel):
        super().__init__()
        self.linear_left = nn.Linear(input_channel, output_channel)
        self.linear_right = nn.Linear(input_channel, output_channel)

    def forward(self, x):
        """x [..., input_channel] -> [..., output_channel] = linear_left(x) * sigmoid(linear_right(x))   (:12-13)."""
        if not x.is_cuda:
            raise _lib.StemGNNHipError(f"input is on {x.device}: stemgnn_amd.GLU runs only on a HIP device (no CPU fallback)")
        lead = x.shape[:-1]
        out = ops.GluFn.apply(x.reshape(-1, x.shape[-1]), self.linear_left.weight, self.linear_left.bias,
                              self.linear_right.weight, self.linear_right.bias)
        return out.reshape(*lead, out.shape[-1])


class StockBlockLayer(nn.Module):
    """Parameters of one spectral block (reference :16-44), created in the reference's order."""

    def __init__(self, time_step, unit, multi_layer, stack_cnt=0):
        super().__init__()
        self.time_step, self.unit, self.multi, self.stack_cnt = time_step, unit, multi_layer, stack_cnt
        wide = time_step * multi_layer
        self.weight = nn.Parameter(torch.empty(1, 4, 1, wide, wide))
        nn.init.xavier_normal_(self.weight)
        self.forecast = nn.Linear(wide, wide)
        self.forecast_result = nn.Linear(wide, time_step)
        if stack_cnt == 0:
            self.backcast = nn.Linear(wide, time_step)
        self.backcast_short_cut = nn.Linear(time_step, time_step)
        self.output_channel = 4 * multi_layer
        glu_out = time_step * self.output_channel
        self.GLUs = nn.ModuleList(
            GLU(time_step * 4 if layer == 0 else glu_out, glu_out) for layer in range(3) for _branch in range(2))

    def hip_params(self):
        """The 33 tensors in the order include/stemgnn_hip.h (SG_BLOCK_NPARAMS) defines; missing backcast -> None."""
        out = [self.weight]
        for name in _BLOCK_FIELDS:
            lin = getattr(self, name, None)
            out += [None, None] if lin is None else [lin.weight, lin.bias]
        for g in self.GLUs:
            out += [g.linear_left.weight, g.linear_left.bias, g.linear_right.weight, g.linear_right.bias]
        return out

    def forward(self, x, mul_L):
        """x [B,1,N,W], mul_L [4,N,N] -> (forecast [B,N,W], backcast [B,1,N,W] | None)   (reference :61-75).
        Stand-alone entry; Model.forward runs both blocks inside one fused autograd node instead."""
        if not x.is_cuda:
            raise _lib.StemGNNHipError(
                f"input is on {x.device}: stemgnn_amd.StockBlockLayer runs only on a HIP device (no CPU fallback)")
        B, _, N, W = x.shape
        forecast, backcast = StockBlockFn.apply(x.reshape(B, N, W), mul_L, self.multi, self.stack_cnt == 0,
                                                *self.hip_params())
        return forecast, (backcast.unsqueeze(1) if backcast is not None else None)


class Model(nn.Module):
    def __init__(self, units, stack_cnt, time_step, multi_layer, horizon=1, dropout_rate=0.5, leaky_rate=0.2,
                 device='cpu'):
        super().__init__()
        # Like the reference, the constructor builds `stack_cnt` blocks for ANY count (:93-95) -- same state_dict -- and it is
        # forward that only works for 2: it sums result[0] + result[1] (:174) and block >= 1 hands None on as the next
        # block's input (:73-75), so 3+ fails on None.unsqueeze and < 2 on result[1]; hot_path reproduces both failures.
        self.unit, self.stack_cnt, self.alpha = units, stack_cnt, leaky_rate
        self.time_step, self.horizon, self.multi_layer = time_step, horizon, multi_layer
        if time_step > _FC_TAIL_MAX_W or horizon > _FC_TAIL_MAX_H:
            # the reference takes any window / horizon through nn.Sequential; the HIP fc tail has a range and there is no
            # torch fallback -- say so at construction (and in README / INTEGRATION), not at the first forward
            raise _lib.StemGNNHipError(
                f"Model(time_step={time_step}, horizon={horizon}): the fc tail kernels (csrc/tail.hip) cover time_step <= "
                f"{_FC_TAIL_MAX_W} and horizon <= {_FC_TAIL_MAX_H}; stemgnn_amd has no torch fallback for larger ones")
        self.dropout_rate = float(dropout_rate)
        self.weight_key = nn.Parameter(torch.zeros(units, 1))
        nn.init.xavier_uniform_(self.weight_key.data, gain=1.414)
        self.weight_query = nn.Parameter(torch.zeros(units, 1))
        nn.init.xavier_uniform_(self.weight_query.data, gain=1.414)
        self.GRU = nn.GRU(time_step, units)
        self.stock_block = nn.ModuleList(
            StockBlockLayer(time_step, units, multi_layer, stack_cnt=i) for i in range(stack_cnt))
        self.fc = nn.Sequential(nn.Linear(time_step, time_step), nn.LeakyReLU(), nn.Linear(time_step, horizon))
        self._seed = None          # device uint64[2] {seed, offset} of the dropout Philox stream (not a parameter)
        self._instance = next(_instance_counter)
        self.hot_state = ops.HotPathState()     # per-model scheduling mode (direct gradients / side-stream overlap)
        self.to(device)

    # -- dropout stream ------------------------------------------------------------------------------
    def _next_seed(self, device):
        if self._seed is None or self._seed.device != device:
            # Philox key of this model's dropout stream.  Like nn.Dropout on a GPU it follows the DEVICE generator's
            # seed (torch.manual_seed sets it) and never consumes torch's CPU generator -- the DataLoader-compatible
            # shuffle order (forecast_dataloader.epoch_order) draws from that one.  The construction index of the model
            # and the data-parallel rank are folded in, so two models / two replicas never share a mask stream.
            rank = 0
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                rank = torch.distributed.get_rank()
            key = (torch.cuda.default_generators[device.index if device.index is not None
                                                 else torch.cuda.current_device()].initial_seed()
                   + 0x9E3779B97F4A7C15 * (self._instance + 1) + 0xD1B54A32D192ED03 * rank) % (1 << 62)
            self._seed = torch.tensor([key, 0], dtype=torch.int64, device=device)
        used = torch.empty_like(self._seed)
        # used := seed, seed.offset += 1 on the device in one launch: graph-capturable, no host sync
        _lib.check(_lib.load().stemgnn_dropout_seed_next(self._seed.data_ptr(), used.data_ptr(),
                                                         torch.cuda.current_stream(device).cuda_stream), "dropout_seed_next")
        return used

    def set_dropout_seed(self, seed, offset=0, device=None):
        device = device or self.weight_key.device
        self._seed = torch.tensor([int(seed), int(offset)], dtype=torch.int64, device=device)

    def __getstate__(self):        # keep whole-module pickling (handler.py:24) working: no streams / device scalars inside
        state = self.__dict__.copy()
        state["_seed"] = None
        state["hot_state"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._instance = next(_instance_counter)
        self.hot_state = ops.HotPathState()

    def train(self, mode=True):
        """nn.Module.train plus the device-side health check of the persistent GRU cluster kernels: the reference driver
        toggles train()/eval() once per epoch / validation pass (models/handler.py:154, :44), so a lost partner workgroup
        (status word set by csrc/gru.hip) raises here instead of training on silently wrong hidden states.  One host
        sync per toggle; skipped while a hipGraph capture is running and before any GRU kernel has run."""
        out = super().train(mode)
        dev = self.weight_key.device if hasattr(self, "weight_key") else None
        if dev is not None and dev.type == "cuda" and ops.gru_status_exists(dev) \
                and not torch.cuda.is_current_stream_capturing():
            ops.check_gru_status(dev)
            if os.environ.get("STEMGNN_SPECTRAL", "cheb") == "eig":
                ops.check_eigh_status(dev)
        return out

    # -- forward --------------------------------------------------------------------------------------
    def prefetch_side(self, device=None, batch=None):
        """Side-stream mode: queue the weight packing and the dropout-stream bookkeeping (a clone and an increment) on the
        side stream NOW, forked from the current stream's position.  hot_path() calls it itself; a step driver
        (engine.TrainStep) calls it before its first kernel of the step, so that the fork does not hang off that kernel:
        inside a hipGraph a node whose successors sit on two queues releases them ~10 us late (measured: window gather ->
        {fill, pack}), and the packing only reads parameters anyway."""
        hs = self.hot_state
        if not hs.overlap or hs.prepacked is not None:
            return
        device = device or self.weight_key.device
        blocks = (self.stock_block[0].hip_params(), self.stock_block[1].hip_params())
        ops.prepack_blocks(hs, blocks, self.time_step, self.multi_layer, device,
                           batch_shape=None if batch is None else (int(batch), self.unit))
        if self.training and self.dropout_rate > 0.0:
            with torch.cuda.stream(ops._side_stream(device)):
                hs.preseed = self._next_seed(device)

    def hot_path(self, x):
        """The persistent HIP GRU recurrence (ops.GruFront; the library GRU only with STEMGNN_GRU=miopen), then the HIP
        hot path; returns (block forecast sum [B,N,W], attention, mul_L)."""
        if not x.is_cuda:
            raise _lib.StemGNNHipError(
                f"input is on {x.device}: stemgnn_amd.Model runs only on a HIP device (no CPU fallback)")
        if self.stack_cnt > 2:        # reference: block 1 returns backcast None (:73-75) -> block 2's x.unsqueeze(1) (:63)
            raise AttributeError("'NoneType' object has no attribute 'unsqueeze'")
        if self.stack_cnt < 2:        # reference: result[0] + result[1] (:174)
            raise IndexError("list index out of range")
        x = x.contiguous()
        blocks = (self.stock_block[0].hip_params(), self.stock_block[1].hip_params())
        use_drop = self.training and self.dropout_rate > 0.0
        hs = self.hot_state
        if hs.overlap and hs.prepacked is None:
            self.prefetch_side(x.device, batch=x.shape[0])
        seed, hs.preseed = hs.preseed, None
        if seed is not None and not torch.cuda.is_current_stream_capturing():
            seed.record_stream(torch.cuda.current_stream())
        if os.environ.get("STEMGNN_GRU", "hip") == "miopen":       # library GRU (MIOpen) -- A/B and debugging only
            h, _ = self.GRU(x.permute(2, 0, 1).contiguous())      # [N_seq, B, N_hid]  (:137)
        else:                                                      # persistent HIP recurrence (csrc/gru.hip)
            g = self.GRU
            h = GruFront.apply(x, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, hs)
            hs.gru_front_live = True       # SpectralHotPath may hand GruFront.backward a factored gradient (ops.py)
        if use_drop and seed is None:
            seed = self._next_seed(x.device)
        params = blocks[0] + blocks[1]
        return SpectralHotPath.apply(h, x, self.weight_key, self.weight_query, self.multi_layer, self.alpha,
                                     self.dropout_rate, self.training, seed, hs, *params)

    def loss(self, x, target, loss_out=None, accum=None, unit_grad=False):
        """MSE training loss of one batch, ``nn.MSELoss()(self(x)[0], target)`` (models/handler.py:161-162), with the fc tail,
        the loss and both their backwards fused into one autograd node (two launches instead of five; forecast itself is
        not materialised).  `loss_out` / `accum`: optional static float32 scalar to write the loss into / float64 scalar
        that receives += loss.  `unit_grad`: the caller promises ``loss.backward()`` with an upstream gradient of 1 (what
        the driver does), which lets direct-gradient mode write the fc gradients in place (only honoured while autograd is
        recording: a logging call under ``torch.no_grad()`` never touches ``p.grad``)."""
        self._require_fc_tail()
        fsum, _attention, _ = self.hot_path(x)
        return FcTailMse.apply(fsum, target, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias,
                               self.hot_state, loss_out, accum, bool(unit_grad) and torch.is_grad_enabled())

    def _require_fc_tail(self):
        """The fc tail kernels (csrc/tail.hip) keep both weight matrices of a row block in LDS: time_step <= 64 and
        horizon <= 32 (every BASELINE configuration; the reference's own runs use 12 / 3).  No torch fallback exists."""
        if not _lib.load().stemgnn_fc_tail_supported(self.time_step, self.horizon):
            raise _lib.StemGNNHipError(f"fc tail: time_step={self.time_step}, horizon={self.horizon} outside the HIP kernels' "
                                       "range (time_step <= 64, horizon <= 32); stemgnn_amd has no torch fallback")

    def forward(self, x):
        self._require_fc_tail()
        fsum, attention, _ = self.hot_path(x)
        # fused fc tail (csrc/tail.hip, models/base_model.py:175-179): Linear - LeakyReLU - Linear and the permute to [B,H,N]
        # in one kernel; for H == 1 the reference's unsqueeze/squeeze (:176-177) yields the same [B,1,N] tensor
        return FcTail.apply(fsum, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias,
                            self.hot_state), attention
