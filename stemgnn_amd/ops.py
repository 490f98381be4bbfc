"""Host-side composition of the HIP stages into one autograd function.

PyTorch is plumbing here: it owns device memory (torch.empty), the current HIP stream and the
autograd graph; every number on the hot path is produced by libstemgnn_hip.so.

Stage order (reference models/base_model.py): attention+Laplacian (:139-147) -> Chebyshev (:148)
-> per StockBlock: pack, GFT (:62-64), spectral GLU (:46-54), IGFT+heads (:55-58, :65-74).
"""
import os
import weakref

import torch

from . import _lib

_NSPLIT = 32      # split-M factor of the slab weight-gradient GEMMs (heads' BS product; GLU only on the fallback path)
_side_streams = {}


def _side_stream(device, index=0):
    key = (str(device), index)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


def fresh_side_stream(device, index=0):
    """Replace the side stream of `device` by a newly created one (engine.TrainStep's schedule self-check: a re-capture
    after a lost branch overlap gives the runtime another stream -> hardware-queue mapping to work with).  The caller
    guarantees that nothing is pending on the old one (device synchronised)."""
    _side_streams.pop((str(device), index), None)
    return _side_stream(device, index)


class HotPathState:
    """Per-model scheduling state of the hot path (one instance per ``stemgnn_amd.Model``; nothing process-global).

    direct : backward WRITES the hot-path / GRU / fc parameter gradients straight into existing ``p.grad`` buffers
             (e.g. the views of a FlatGradBucket) instead of returning them to autograd, which would launch one
             accumulate kernel per parameter (~70 tiny launches per step).  Semantics: overwrite, i.e. equivalent to
             ``zero_grad(); backward()`` -- not for accumulating gradients over several backward passes.
    overlap: additionally queue the spectral blocks' weight-gradient GEMMs (and the weight packing / dropout-stream
             bookkeeping of the forward) on a side HIP stream so they overlap the rest of the step (notably the
             latency-bound GRU recurrence).  Gradients are then complete only after ``join_side_streams()``;
             FusedRMSprop.step, FlatGradBucket.all_reduce_mean and GruFront.backward call it -- any other reader of
             ``.grad`` must call it first.
    """

    def __init__(self):
        self.direct = False
        self.overlap = False
        self.prepacked = None        # (packed panels, side stream, blocks) queued by prepack_blocks
        self.preseed = None          # dropout key of the coming forward, cloned on the side stream (Model.prefetch_side)
        self.fork_event = None       # optional event the next prepack_blocks forks the side stream from
        self.pending = None          # (side stream, keep-alive objects) of weight-gradient work not yet joined
        self.exact_group = None      # data-parallel "exact mode" (SURVEY 8e-ii): (process group, world size) or None
        self.gru_front_live = False  # set by Model.hot_path when the GRU output it hands to SpectralHotPath came from GruFront
        self.dh_factors = None       # (attn scratch, B, N, wk, wq): the factored d(loss)/d(GRU output) SpectralHotPath.backward left
                                     # for GruFront.backward (stemgnn_gru_bwd_rank2) instead of a materialised [N,B,N] tensor
        self.block_grads_hook = None # callable run on the side stream right behind block 1's un-packing (overlap mode): the
                                     # step driver's all-reduce of the block / fc gradient range (engine.TrainStep)
        self.gru_ctl = None          # int32 control words of the GRU's dW_hh product beside the recurrence (stemgnn_gru_bwd_rank2_begin
        self.gru_ctl_zeroed = False  # / _finish); True: zeroed on the side stream by this backward pass, ahead of both streams' use
        self.warm_saved = None       # dummy saved-activation buffer of the fused forward's warm-up launch (prepack_blocks)
        self.tail_finish = None      # thunk(stream): the fc tail's partial-sum launch (loss, fc gradients) FcTailMse.forward left for
                                     # SpectralHotPath.backward to queue on the side branch (nothing on the chain reads its output)
        self.side_probe = None       # a list: every kernel the step would put on the SIDE branch is also appended as a
                                     # re-issuable thunk(stream) -- engine.TrainStep's schedule self-check replays them alone to
                                     # measure the side branch's kernel-time sum (collectives and the dropout key step excluded)
        _states.add(self)

    def set(self, direct=True, overlap=False):
        self.direct = bool(direct)
        self.overlap = bool(direct) and bool(overlap)
        return self

    def join(self):
        item, self.pending = self.pending, None
        if item is not None:
            side, keep = item
            torch.cuda.current_stream().wait_stream(side)
            del keep

    def reset(self):
        """Drop queued side-stream work after a failed capture / aborted step (engine.capture)."""
        for item in (self.prepacked, self.pending):
            if item is not None:
                try:        # the stream may have been forked into the capture that just failed (invalidated capture)
                    (item[1] if item is self.prepacked else item[0]).synchronize()
                except RuntimeError:
                    pass    # engine.capture synchronizes the device afterwards anyway
        self.prepacked = None
        self.preseed = None
        self.fork_event = None
        self.pending = None
        self.dh_factors = None
        self.gru_ctl_zeroed = False
        self.tail_finish = None


_states = weakref.WeakSet()
_NO_STATE = None


def _state(state):
    """Functions called without a model (stand-alone use, tests): a shared inert state (no direct writes, no overlap)."""
    global _NO_STATE
    if state is not None:
        return state
    if _NO_STATE is None:
        _NO_STATE = HotPathState()
    return _NO_STATE


def join_side_streams(device=None):
    """Make the current stream wait for weight-gradient work any model queued on a side stream (see
    SpectralHotPath.backward).  Called by every consumer of the gradients; cheap no-op when nothing is pending."""
    for st in list(_states):
        if st.pending is not None and (device is None or str(st.pending[0].device) == str(torch.device(device))):
            st.join()


def _all_reduce_mean(t, group_world):
    """Mean of `t` over the ranks of (group, world[, stand-in]) on the current stream.  The optional third entry is an
    in-stream stand-in fn(view) for the SUM all-reduce (schedule tests on a 1-GPU box, see FlatGradBucket.reduce_fn)."""
    group, world = group_world[0], group_world[1]
    fn = group_world[2] if len(group_world) > 2 else None
    if fn is not None:
        fn(t)
    else:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.div_(world)


def set_direct_grad(model, flag=True, overlap=False):
    """Switch ``model`` (a stemgnn_amd.Model) to direct-gradient mode, see HotPathState."""
    return model.hot_state.set(flag, overlap)


_NCHUNK = 16      # row chunks of the attention backward


def _wg_cu(which, B, N):
    """CU share (percent) of the side stream's first / second fused weight-gradient launch.  The first runs beside the
    Chebyshev / attention backward chain and is sized for (nearly) the whole chip (round 4: 80 / 65 % measured 0 / +18 us per
    step; round 5: see below); the second
    runs under the GRU backward recurrence, which pins stemgnn_gru_bwd_cus(B, N) CUs for its whole run, and is sized for
    what is left (50 % at PEMS07 where 4 workgroups serve a batch row, 25 % at N = 358 with 6; hidden sizes of the wide
    cluster keep 50)."""
    if which == 0:
        # round 5, with the 256 x 64 tiles (33 instead of 37 tiles per launch): 100 % would give 7 splits = 231 workgroups and
        # starves the backward chain beside it (ChebBwd 17 + 21 -> 30 + 42 us; step 1.142 ms against 1.122 at 90 % = 6 splits,
        # 198 workgroups; 80 % 1.122; the 128 x 128-only build 1.125 -- same box, profiles/r05_wgrad_tiles_ab.txt)
        return 90
    if N > 512:          # wide cluster (hidden > 512): the weight-gradient work there exceeds what the idle CUs could do in
        return 50        # the recurrence's time -- the round-2/3 setting (half of the chip, sharing CUs with the cluster) stays
    lib = _lib.load()
    cus = lib.stemgnn_num_cus()
    busy = lib.stemgnn_gru_bwd_cus(B, N)
    return max(10, min(100, (cus - busy) * 100 // cus))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def glu_splits():
    """Arithmetic of the GLU forward / data-gradient layers inside the model (STEMGNN_DTYPE): 0 = exact fp32 MFMA (default,
    the reference's arithmetic), 3 / 2 = split-bf16 products (bf16x3: fp32-class; bf16x2: ~2^-16 relative) on the bf16
    matrix instructions with fp32 accumulation (csrc/gemm2s.h).  Read per call, so a test can switch it."""
    v = os.environ.get("STEMGNN_DTYPE", "f32").lower()
    if v in ("f32", "fp32", ""):
        return 0
    if v == "bf16x3":
        return 3
    if v == "bf16x2":
        return 2
    raise _lib.StemGNNHipError(f"STEMGNN_DTYPE={v!r}: expected f32, bf16x3 or bf16x2")


def _pack_block(lib, parr, tables, pk, W, multi, splits, stream):
    """Pack one block's weights for the arithmetic the GLU products will run in: exact fp32 -> panels + the fp32 stage streams of
    the fused kernels (stemgnn_block_pack); split-bf16 -> the panels only (stemgnn_block_pack_panels; _split_panels packs the
    bf16 streams from them and nothing reads the fp32 streams: two dead launches and ~35 MB of writes per step in round 5)."""
    if splits:
        _lib.check(lib.stemgnn_block_pack_panels(parr, tables.data_ptr(), pk.data_ptr(), W, multi, stream), "block_pack_panels")
    else:
        _lib.check(lib.stemgnn_block_pack(parr, tables.data_ptr(), pk.data_ptr(), W, multi, stream), "block_pack")


def _split_panels(lib, pk, W, multi, splits, device, stream):
    sp = torch.empty(lib.stemgnn_glu_split_floats(W, multi, splits), device=device, dtype=torch.float32)
    _lib.check(lib.stemgnn_glu_split_panels(pk.data_ptr(), sp.data_ptr(), W, multi, splits, stream), "glu_split_panels")
    return sp


def _require_gpu(t, name):
    if not t.is_cuda:
        raise _lib.StemGNNHipError(
            f"{name} is on {t.device}: stemgnn_amd runs only on a HIP device (no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.StemGNNHipError(f"{name} must be float32, got {t.dtype}")


_table_cache = {}


def dft_tables(W, multi, device):
    """Constant DFT / C2R tables for (W, multi), built once on the host by the library."""
    key = (W, multi, str(device))
    t = _table_cache.get(key)
    if t is None:
        lib = _lib.load()
        n = lib.stemgnn_table_floats(W, multi)
        host = torch.empty(n, dtype=torch.float32)
        _lib.check(lib.stemgnn_make_tables_host(W, multi, host.data_ptr()), "make_tables_host")
        t = host.to(device)
        _table_cache[key] = t
    return t


def dropout_mask(drop_p, seed, B, N):
    """Test hook: the 0/1 keep mask [B,N,N] the attention kernels regenerate from `seed` (uint64[2])."""
    lib = _lib.load()
    mask = torch.empty(B, N, N, device=seed.device, dtype=torch.float32)
    _lib.check(lib.stemgnn_dropout_mask(float(drop_p), seed.data_ptr(), B, N, mask.data_ptr(), _stream()),
               "dropout_mask")
    return mask


_gru_status = {}


def gru_status(device):
    """Device int32 the cluster GRU kernels set to 1 if an inter-workgroup wait timed out (never expected)."""
    key = str(device)
    t = _gru_status.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _gru_status[key] = t
    return t


def gru_status_exists(device):
    """True once a GRU kernel has been launched on `device` (the status word is created by the first launch)."""
    return str(device) in _gru_status


def check_eigh_status(device=None):
    """Host sync + raise if the direct eigensolver flagged a run on `device` (STEMGNN_SPECTRAL=eig only).  The library
    keeps one status word per device and clears it when it is read, so an event is reported once."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    with torch.cuda.device(dev):
        rc = _lib.load().stemgnn_eigh_status()
    if rc == 2:
        raise _lib.StemGNNHipError(f"eigensolver status 2 on {dev}: a grid-barrier wait of the tridiagonalisation timed "
                                   "out (a workgroup of its cluster was not resident)")
    if rc == 3:
        raise _lib.StemGNNHipError(f"eigensolver status 3 on {dev}: the re-solve of an eigenvalue cluster broke down "
                                   "(no independent start vector left)")
    if rc != 0:
        raise _lib.StemGNNHipError(f"eigensolver status {rc} on {dev}")


def check_gru_status(device):
    """Host sync + raise if a GRU exchange timed out (call outside timed regions / in tests)."""
    st = gru_status(device)
    if int(st.item()) != 0:
        st.zero_()                               # report once; the next check sees only new time-outs
        raise _lib.StemGNNHipError("GRU cluster exchange timed out (partner workgroup not resident?); "
                                   "set STEMGNN_GRU_CLUSTER=0 to use the single-workgroup kernels")


class GruFront(torch.autograd.Function):
    """nn.GRU(time_step, units) over the node axis (reference models/base_model.py:137) as two persistent HIP
    recurrence kernels.  (x [B,W,N], weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0) -> h [N,B,N]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, state=None):
        lib = _lib.load()
        ctx.state = _state(state)
        gru_params = (w_ih, w_hh, b_ih, b_hh)
        for name, t in (("x", x), ("GRU.weight_ih_l0", w_ih), ("GRU.weight_hh_l0", w_hh)):
            _require_gpu(t, name)
        x = x.contiguous()
        B, W, S = x.shape
        Hd = w_hh.shape[1]
        dev, f32 = x.device, torch.float32
        w_ih, w_hh, b_ih, b_hh = (t.contiguous() for t in (w_ih, w_hh, b_ih, b_hh))
        h_ext = torch.empty(S + 1, B, Hd, device=dev, dtype=f32)     # slab 0 = h_{-1} = 0 (set by the library)
        h_all = h_ext[1:]                                             # exactly nn.GRU's output, a contiguous view
        reserve = torch.empty(lib.stemgnn_gru_reserve_floats(B, S, Hd), device=dev, dtype=f32)
        scratch = torch.empty(lib.stemgnn_gru_fwd_scratch_floats(B, S, Hd), device=dev, dtype=f32)
        _lib.check(lib.stemgnn_gru_fwd(x.data_ptr(), w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(),
                                       B, S, Hd, W, scratch.data_ptr(), h_ext.data_ptr(), reserve.data_ptr(),
                                       gru_status(dev).data_ptr(), _stream()), "gru_fwd")
        # save_for_backward (not ctx attributes): h_all is an OUTPUT -- holding it on ctx would form a
        # ctx <-> grad_fn reference cycle that never frees the step's buffers
        ctx.save_for_backward(x, w_ih, w_hh, h_ext, reserve)
        ctx.gru_params = gru_params
        return h_all

    @staticmethod
    def backward(ctx, dh_all):
        lib = _lib.load()
        x, w_ih, w_hh, h_ext, reserve = ctx.saved_tensors
        B, W, S = x.shape
        Hd = w_hh.shape[1]
        dev, f32 = x.device, torch.float32
        factors, ctx.state.dh_factors = ctx.state.dh_factors, None
        if factors is None:
            dh_all = dh_all.contiguous()
        scratch = torch.empty(lib.stemgnn_gru_bwd_scratch_floats(B, S, Hd, W), device=dev, dtype=f32)
        prm = ctx.gru_params
        direct = ctx.state.direct and all(p.grad is not None and p.grad.is_contiguous() for p in prm)
        if direct:
            dw_ih, dw_hh, db_ih, db_hh = (p.grad for p in prm)
        else:
            dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
            db_ih = torch.empty(3 * Hd, device=dev, dtype=f32)
            db_hh = torch.empty(3 * Hd, device=dev, dtype=f32)
        if factors is not None:
            # SpectralHotPath.backward stopped at dkey | dquery: dh[s,b,i] = dkey[b,i] wk[s] + dquery[b,i] wq[s] is formed
            # inside the recurrence (the [N,B,N] gradient tensor is never written; `dh_all` is a zero-stride placeholder)
            attn_scratch, fb, fn, wk, wq, after, dq_chunks = factors
            base = attn_scratch.data_ptr() + 4 * fn * fn
            tail = (x.data_ptr(), w_hh.data_ptr(), h_ext.data_ptr(), reserve.data_ptr(), B, S, Hd, W, scratch.data_ptr(),
                    dw_ih.data_ptr(), dw_hh.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(), gru_status(dev).data_ptr())
            args = (base, base + 4 * fb * fn, wk.data_ptr(), wq.data_ptr()) + tail
            # overlap mode: the dW_hh product runs on the side stream BESIDE the recurrence (persistent workgroups that follow
            # its progress counters) and only the last few time steps' share + the fixed-order sums stay behind it
            # (include/stemgnn_hip.h: stemgnn_gru_bwd_rank2_begin / _finish; same bits as the single call)
            beside = (ctx.state.overlap and ctx.state.pending is not None and ctx.state.gru_ctl_zeroed
                      and bool(lib.stemgnn_gru_bwd_overlap_ok(B, S, Hd, W))
                      and ctx.state.gru_ctl.numel() == lib.stemgnn_gru_bwd_ctl_words(S))
            ctx.state.gru_ctl_zeroed = False
            ctl = ctx.state.gru_ctl.data_ptr() if beside else None
            if beside:
                _lib.check(lib.stemgnn_gru_bwd_rank2_begin(*args, ctl, _stream()), "gru_bwd_rank2_begin")
            elif dq_chunks:
                # dquery is still the attention backward's per-chunk partials: summed inside this call's zero-fill launch
                # (bf16x2: the dW_hh product on the bf16 matrix pipe as well, like the blocks' weight gradients)
                wg_split = 1 if glu_splits() == 2 and os.environ.get("STEMGNN_WGRAD_BF16", "1") != "0" else 0
                _lib.check(lib.stemgnn_gru_bwd_rank2_dq(base, base + 4 * fb * fn, dq_chunks, wg_split, wk.data_ptr(), wq.data_ptr(),
                                                        *tail, _stream()), "gru_bwd_rank2_dq")
            else:
                _lib.check(lib.stemgnn_gru_bwd_rank2(*args, _stream()), "gru_bwd_rank2")
            if after is not None:
                after()                 # dwk / dwq on the side stream (needs dkey | dquery only)
            if beside:
                # behind dwk / dwq on the side stream (its only parent inside a captured graph: a parent on the main branch
                # would make it wait for the recurrence itself, include/stemgnn_hip.h).  A third stream that starts the
                # followers WITH the recurrence was measured too: they take CUs from block 1's weight gradients, which then
                # end behind the recurrence (+26 us per step; profiles/r05_gru_whh_overlap.md)
                side = ctx.state.pending[0]
                _lib.check(lib.stemgnn_gru_bwd_rank2_finish(*args, ctl, side.cuda_stream, _stream()), "gru_bwd_rank2_finish")
        else:
            _lib.check(lib.stemgnn_gru_bwd(dh_all.data_ptr(), x.data_ptr(), w_hh.data_ptr(), h_ext.data_ptr(),
                                           reserve.data_ptr(), B, S, Hd, W, scratch.data_ptr(), dw_ih.data_ptr(),
                                           dw_hh.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr(),
                                           gru_status(dev).data_ptr(), _stream()), "gru_bwd")
        ctx.state.join()                # the spectral blocks' weight gradients (side stream) overlapped this recurrence
        if direct:
            return None, None, None, None, None, None
        return None, dw_ih, dw_hh, db_ih, db_hh, None


class FcTail(torch.autograd.Function):
    """fc tail of Model.forward (reference models/base_model.py:175-179): fsum [B,N,W] -> forecast [B,H,N]."""

    @staticmethod
    def forward(ctx, fsum, w0, b0, w2, b2, state=None):
        lib = _lib.load()
        ctx.state = _state(state)
        fsum = fsum.contiguous()
        B, N, W = fsum.shape
        H = w2.shape[0]
        forecast = torch.empty(B, H, N, device=fsum.device, dtype=torch.float32)
        w0, b0, w2, b2 = (t.contiguous() for t in (w0, b0, w2, b2))
        _lib.check(lib.stemgnn_fc_tail_fwd(fsum.data_ptr(), w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                           B, N, W, H, forecast.data_ptr(), _stream()), "fc_tail_fwd")
        ctx.save_for_backward(fsum, w0, b0, w2)
        ctx.fc_params = (w0, b0, w2, b2)
        return forecast

    @staticmethod
    def backward(ctx, dforecast):
        lib = _lib.load()
        fsum, w0, b0, w2 = ctx.saved_tensors
        B, N, W = fsum.shape
        H = w2.shape[0]
        dev, f32 = fsum.device, torch.float32
        dforecast = dforecast.contiguous()
        prm = ctx.fc_params
        direct = ctx.state.direct and all(p.grad is not None and p.grad.is_contiguous() for p in prm)
        if direct:
            dw0, db0, dw2, db2 = (p.grad for p in prm)
        else:
            dw0, db0, dw2, db2 = (torch.empty_like(p) for p in prm)
        dfsum = torch.empty_like(fsum)
        scratch = torch.empty(lib.stemgnn_fc_tail_scratch_floats(B, N, W, H), device=dev, dtype=f32)
        _lib.check(lib.stemgnn_fc_tail_bwd(dforecast.data_ptr(), fsum.data_ptr(), w0.data_ptr(), b0.data_ptr(),
                                           w2.data_ptr(), B, N, W, H, scratch.data_ptr(), dfsum.data_ptr(),
                                           dw0.data_ptr(), db0.data_ptr(), dw2.data_ptr(), db2.data_ptr(), _stream()),
                   "fc_tail_bwd")
        if direct:
            return dfsum, None, None, None, None, None
        return dfsum, dw0, db0, dw2, db2, None


class FcTailMse(torch.autograd.Function):
    """Training tail as one node: fc tail (reference models/base_model.py:175-179) + nn.MSELoss(reduction='mean')
    (models/handler.py:140,162) + both backwards, two launches (``stemgnn_fc_tail_train``).  forward returns the loss
    and already holds d(loss)/d(fsum) and the fc parameter gradients for an upstream gradient of 1 (``loss.backward()``,
    handler.py:164); backward hands them out (scaled by the upstream gradient unless ``unit_grad``).
    (fsum [B,N,W], target [B,H,N], fc.0.weight, fc.0.bias, fc.2.weight, fc.2.bias) -> loss []."""

    @staticmethod
    def forward(ctx, fsum, target, w0, b0, w2, b2, state=None, loss_out=None, accum=None, unit_grad=False):
        lib = _lib.load()
        ctx.state = _state(state)
        _require_gpu(fsum, "fsum")
        _require_gpu(target, "target")
        fsum, target = fsum.contiguous(), target.contiguous()
        B, N, W = fsum.shape
        H = w2.shape[0]
        if tuple(target.shape) != (B, H, N):
            raise _lib.StemGNNHipError(f"target must be [B,H,N]=({B},{H},{N}), got {tuple(target.shape)}")
        dev, f32 = fsum.device, torch.float32
        prm = (w0, b0, w2, b2)
        w0c, b0c, w2c, b2c = (t.contiguous() for t in prm)
        # direct writes into p.grad only for a call that WILL be back-propagated with an upstream gradient of 1
        # (unit_grad: the step driver's promise): a logging call under no_grad, or a scaled loss, must not clobber the
        # flat gradient bucket -- those use temporaries that backward scales and returns
        wanted = any(ctx.needs_input_grad[i] for i in (0, 2, 3, 4, 5))
        direct = ctx.state.direct and bool(unit_grad) and wanted and \
            all(p.grad is not None and p.grad.is_contiguous() for p in prm)
        grads = [p.grad for p in prm] if direct else [torch.empty_like(p) for p in prm]
        dfsum = torch.empty_like(fsum)
        loss = loss_out if loss_out is not None else torch.empty((), device=dev, dtype=f32)
        if accum is not None and (accum.dtype != torch.float64 or accum.device != dev):
            raise _lib.StemGNNHipError("accum must be a float64 scalar on the forecast's device")
        scratch = torch.empty(lib.stemgnn_fc_tail_train_scratch_floats(B, N, W, H), device=dev, dtype=f32)

        acc_ptr = accum.data_ptr() if accum is not None else None
        # direct + unit_grad + side-stream mode: the step driver promises an immediate backward with an upstream gradient of 1
        # (engine.TrainStep) -> only the per-row launch runs here; the partial-sum launch (loss, fc gradients: no consumer
        # before the optimizer) is left for SpectralHotPath.backward to queue on the side branch (-10 us on the chain)
        defer = direct and ctx.state.overlap and ctx.state.tail_finish is None
        if defer:
            _lib.check(lib.stemgnn_fc_tail_train_rows(
                fsum.data_ptr(), target.data_ptr(), w0c.data_ptr(), b0c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), B, N, W, H,
                scratch.data_ptr(), None, dfsum.data_ptr(), _stream()), "fc_tail_train_rows")
            g0, g1, g2, g3 = grads

            def finish(stream, scratch=scratch, loss=loss, g0=g0, g1=g1, g2=g2, g3=g3):
                _lib.check(lib.stemgnn_fc_tail_train_finish(scratch.data_ptr(), B, N, W, H, loss.data_ptr(), acc_ptr, g0.data_ptr(),
                                                            g1.data_ptr(), g2.data_ptr(), g3.data_ptr(), stream),
                           "fc_tail_train_finish")
            ctx.state.tail_finish = finish
        else:
            _lib.check(lib.stemgnn_fc_tail_train(
                fsum.data_ptr(), target.data_ptr(), w0c.data_ptr(), b0c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(), B, N, W, H,
                scratch.data_ptr(), None, loss.data_ptr(), acc_ptr, dfsum.data_ptr(),
                grads[0].data_ptr(), grads[1].data_ptr(), grads[2].data_ptr(), grads[3].data_ptr(), _stream()), "fc_tail_train")
        # (the reduction of the per-block partials -- loss, fc gradients -- stays a launch of its own on this stream.  Measured
        # in round 4: on the side stream no gain (the cross-queue edge costs what the 9 us launch returns); as the job of the
        # LAST workgroup to arrive of the first launch, 40 us instead of 12 + 10 -- one workgroup summing 114 partials that
        # other XCDs just wrote is a serial chain of fabric round trips)
        ctx.direct, ctx.unit_grad = direct, bool(unit_grad)
        ctx.held = (dfsum, None if direct else grads)
        ctx.set_materialize_grads(False)
        if loss_out is not None:
            ctx.mark_dirty(loss_out)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        dfsum, grads = ctx.held
        ctx.held = None
        if ctx.state.tail_finish is not None and not ctx.state.overlap:      # the mode changed between forward and backward
            fin, ctx.state.tail_finish = ctx.state.tail_finish, None
            fin(_stream())
        if not ctx.unit_grad:
            dfsum = dfsum * grad_loss
            if grads is not None:
                grads = [g * grad_loss for g in grads]
        if grads is None:
            return dfsum, None, None, None, None, None, None, None, None, None
        return dfsum, None, grads[0], grads[1], grads[2], grads[3], None, None, None, None


class GluFn(torch.autograd.Function):
    """Stand-alone GLU (reference models/base_model.py:6-13): x [M,K] -> (x Wl^T + bl) * sigmoid(x Wr^T + br) [M,C].
    Inside the model the GLU is the epilogue of the fused spectral GEMMs; this composition (general fp32 GEMM entry +
    elementwise kernels) only backs ``stemgnn_amd.GLU.forward`` for callers that use the module on its own."""

    @staticmethod
    def forward(ctx, x, wl, bl, wr, br):
        lib = _lib.load()
        for name, t in (("x", x), ("linear_left.weight", wl), ("linear_right.weight", wr)):
            _require_gpu(t, name)
        x, wl, bl, wr, br = (t.contiguous() for t in (x, wl, bl, wr, br))
        M, K = x.shape
        C = wl.shape[0]
        dev, st = x.device, _stream()
        U = torch.empty(M, C, device=dev)
        V = torch.empty(M, C, device=dev)
        for w, o in ((wl, U), (wr, V)):                 # x W^T: both operands k-contiguous
            _lib.check(lib.stemgnn_sgemm_f32(x.data_ptr(), K, 1, w.data_ptr(), K, 1, o.data_ptr(), C, M, C, K, 0, st), "sgemm")
        out, gate, lin = (torch.empty(M, C, device=dev) for _ in range(3))
        _lib.check(lib.stemgnn_glu_combine_fwd(U.data_ptr(), V.data_ptr(), bl.data_ptr(), br.data_ptr(), out.data_ptr(),
                                               gate.data_ptr(), lin.data_ptr(), M, C, st), "glu_combine_fwd")
        ctx.save_for_backward(x, wl, wr, lin, gate)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, wl, wr, lin, gate = ctx.saved_tensors
        M, K = x.shape
        C = wl.shape[0]
        dev, st = x.device, _stream()
        dout = dout.contiguous()
        dU, dV = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)
        _lib.check(lib.stemgnn_glu_combine_bwd(dout.data_ptr(), lin.data_ptr(), gate.data_ptr(), dU.data_ptr(),
                                               dV.data_ptr(), M, C, st), "glu_combine_bwd")
        dx = torch.empty(M, K, device=dev)
        dwl, dwr = torch.empty_like(wl), torch.empty_like(wr)
        dbl, dbr = torch.empty(C, device=dev), torch.empty(C, device=dev)
        for i, (d, w, dw, db) in enumerate(((dU, wl, dwl, dbl), (dV, wr, dwr, dbr))):
            # dx (+)= d W : A = d [M,C] k-contiguous, B(k=c, j) = W[c*K + j] (j-contiguous)
            _lib.check(lib.stemgnn_sgemm_f32(d.data_ptr(), C, 1, w.data_ptr(), K, 0, dx.data_ptr(), K, M, K, C, int(i > 0), st), "sgemm")
            # dW = d^T x : A(i=c, k=m) = d[m*C + c], B(k=m, j) = x[m*K + j]
            _lib.check(lib.stemgnn_sgemm_f32(d.data_ptr(), C, 0, x.data_ptr(), K, 0, dw.data_ptr(), K, C, K, M, 0, st), "sgemm")
            _lib.check(lib.stemgnn_colsum(d.data_ptr(), M, C, db.data_ptr(), st), "colsum")
        return dx, dwl, dbl, dwr, dbr


class StockBlockFn(torch.autograd.Function):
    """One StockBlockLayer (reference models/base_model.py:61-75) as a stand-alone autograd node:
    (X [B,N,W] contiguous, mul_L [4,N,N], multi, has_backcast, 33 block params) -> (forecast [B,N,W], backcast [B,N,W]).
    Model.forward uses the fused two-block node (SpectralHotPath); this one backs StockBlockLayer.forward."""

    @staticmethod
    def forward(ctx, X, mul_L, multi, has_bc, *params):
        lib = _lib.load()
        _require_gpu(X, "x")
        _require_gpu(mul_L, "mul_L")
        X = X.contiguous()
        mul_L = mul_L.contiguous()
        B, N, W = X.shape
        dev, f32 = X.device, torch.float32
        st = _stream()
        params = [None if p is None else p.contiguous() for p in params]
        tables = dft_tables(W, multi, dev)
        pk = torch.empty(lib.stemgnn_packed_floats(W, multi), device=dev, dtype=f32)
        sv = torch.empty(lib.stemgnn_saved_floats(B, N, W, multi), device=dev, dtype=f32)
        forecast = torch.empty(B, N, W, device=dev, dtype=f32)
        backcast = torch.empty(B, N, W, device=dev, dtype=f32) if has_bc else None
        parr = _lib.ptr_array(params)
        sb, sn, stt = N * W, W, 1
        _lib.check(lib.stemgnn_block_pack(parr, tables.data_ptr(), pk.data_ptr(), W, multi, st), "block_pack")
        _lib.check(lib.stemgnn_gft_fwd(mul_L.data_ptr(), X.data_ptr(), sb, sn, stt, sv.data_ptr(), B, N, W, st), "gft_fwd")
        _lib.check(lib.stemgnn_spectral_glu_fwd(pk.data_ptr(), sv.data_ptr(), B, N, W, multi, st), "spectral_glu_fwd")
        _lib.check(lib.stemgnn_igft_heads_fwd(parr, pk.data_ptr(), sv.data_ptr(), X.data_ptr(), sb, sn, stt,
                                              forecast.data_ptr(), 0, backcast.data_ptr() if has_bc else None,
                                              B, N, W, multi, st), "igft_heads_fwd")
        ctx.dims = (B, N, W, multi, bool(has_bc))
        ctx.params = params
        ctx.aux = (X, mul_L, tables, pk, sv, backcast.detach() if has_bc else None)
        if has_bc:
            return forecast, backcast
        return forecast, None

    @staticmethod
    def backward(ctx, dforecast, dbackcast):
        lib = _lib.load()
        B, N, W, multi, has_bc = ctx.dims
        X, mul_L, tables, pk, sv, backcast = ctx.aux
        params = ctx.params
        dev, f32 = X.device, torch.float32
        st = _stream()
        nsplit = _NSPLIT
        if dforecast is None:
            dforecast = torch.zeros(B, N, W, device=dev, dtype=f32)
        dforecast = dforecast.contiguous()
        use_bc = has_bc and dbackcast is not None
        if has_bc and not use_bc:
            dbackcast = torch.zeros(B, N, W, device=dev, dtype=f32)
            use_bc = True
        scratch = torch.empty(lib.stemgnn_scratch_floats(B, N, W, multi), device=dev, dtype=f32)
        dG = scratch[lib.stemgnn_scratch_offset_dG(B, N, W, multi):]
        gradpart = torch.empty(lib.stemgnn_gradpart_floats(W, multi, nsplit), device=dev, dtype=f32)
        dmul_L = torch.zeros(4, N, N, device=dev, dtype=f32)
        dX = torch.empty(B, N, W, device=dev, dtype=f32)
        parr = _lib.ptr_array(params)
        sb, sn, stt = N * W, W, 1
        _lib.check(lib.stemgnn_igft_heads_bwd(
            parr, pk.data_ptr(), sv.data_ptr(), X.data_ptr(), sb, sn, stt, dforecast.data_ptr(),
            dbackcast.contiguous().data_ptr() if use_bc else None, backcast.data_ptr() if use_bc else None,
            scratch.data_ptr(), gradpart.data_ptr(), nsplit, 1, B, N, W, multi, st), "igft_heads_bwd")
        _lib.check(lib.stemgnn_spectral_glu_bwd(pk.data_ptr(), sv.data_ptr(), scratch.data_ptr(), gradpart.data_ptr(),
                                                nsplit, 1, B, N, W, multi, st), "spectral_glu_bwd")
        _lib.check(lib.stemgnn_block_wgrad(parr, pk.data_ptr(), sv.data_ptr(), X.data_ptr(), sb, sn, stt,
                                           dforecast.data_ptr(), int(use_bc), scratch.data_ptr(), gradpart.data_ptr(),
                                           nsplit, 100, B, N, W, multi, st), "block_wgrad")
        _lib.check(lib.stemgnn_gft_bwd(mul_L.data_ptr(), X.data_ptr(), sb, sn, stt, dG.data_ptr(), dX.data_ptr(),
                                       dmul_L.data_ptr(), 0, B, N, W, st), "gft_bwd")
        grads = [None] * 33
        for i, p in enumerate(params):
            if p is None or (not has_bc and i in (7, 8)):
                continue
            grads[i] = torch.empty_like(p)
        _lib.check(lib.stemgnn_block_unpack_grads(gradpart.data_ptr(), nsplit, tables.data_ptr(), _lib.ptr_array(grads),
                                                  W, multi, int(has_bc), st), "block_unpack_grads")
        if use_bc and ctx.needs_input_grad[0]:
            # short-cut input of the backcast head (:71-72): backcast = sigmoid(BC(ig) - BS(x)) also depends on x directly.
            # Model.forward never needs this term (x is data); only a stand-alone block with a differentiable input does.
            _lib.check(lib.stemgnn_shortcut_dx(scratch.data_ptr(), params[7].data_ptr(), dX.data_ptr(), B, N, W, multi, st),
                       "shortcut_dx")
        return (dX, dmul_L, None, None, *grads)


_keep_attention_state = False
_last_attention_state = {}


def capture_attention_state(flag=True):
    """Test hook: keep the key / query vectors ([B,N] each) of the last SpectralHotPath.forward per device, so a
    reference computed at higher precision can take the same LeakyReLU-kink decisions (key_i + query_j > 0)."""
    global _keep_attention_state
    _keep_attention_state = bool(flag)
    if not flag:
        _last_attention_state.clear()


def last_attention_state(device):
    return _last_attention_state.get(str(torch.device(device)))


def prepack_blocks(state, block_params, W, multi, device, batch_shape=None):
    """Pack both blocks' weights (stemgnn_block_pack) on the side stream now, so that they overlap whatever the
    caller queues next on the current stream (Model.hot_path: the GRU recurrence).  The next SpectralHotPath.forward
    on this device picks the packed panels up and joins the side stream.  batch_shape = (B, N) of the coming forward, when the
    caller knows it: the fused forward is warmed for that shape behind the packing."""
    lib = _lib.load()
    side, main = _side_stream(device), torch.cuda.current_stream()
    tables = dft_tables(W, multi, device)
    n_packed = lib.stemgnn_packed_floats(W, multi)
    blocks = [[None if p is None else p.contiguous() for p in blk] for blk in block_params]
    packed = [torch.empty(n_packed, device=device, dtype=torch.float32) for _ in blocks]
    ev, state.fork_event = state.fork_event, None
    if ev is not None:
        side.wait_event(ev)          # fork point chosen by the step driver (engine.TrainStep: ahead of the window gather)
    else:
        side.wait_stream(main)
    splits = glu_splits()
    split = []
    for blk, pk in zip(blocks, packed):
        parr = _lib.ptr_array(blk)

        def pack(stream, parr=parr, pk=pk, blk=blk):
            _pack_block(lib, parr, tables, pk, W, multi, splits, stream)
        pack(side.cuda_stream)
        if state.side_probe is not None:
            state.side_probe.append(pack)
        if splits:      # (allocated on the current stream like the packed panels; the side stream only runs the kernel)
            split.append(_split_panels(lib, pk, W, multi, splits, device, side.cuda_stream))
    # control words of the GRU's dW_hh product beside the backward recurrence (opt-in, STEMGNN_GRU_WHH_OVERLAP=1): zeroed HERE,
    # on the side branch under the GRU forward -- ordered ahead of the recurrence through the join that precedes the attention
    # kernels, and ahead of the side launch that reads them by stream order (round 6: the backward's schedule has no side ->
    # main edge ahead of the recurrence any more)
    if batch_shape is not None and bool(lib.stemgnn_gru_bwd_overlap_ok(batch_shape[0], batch_shape[1], batch_shape[1], W)):
        nctl = lib.stemgnn_gru_bwd_ctl_words(batch_shape[1])
        if state.gru_ctl is None or state.gru_ctl.numel() != nctl or state.gru_ctl.device != torch.device(device):
            state.gru_ctl = torch.zeros(nctl, device=device, dtype=torch.int32)
        _lib.check(lib.stemgnn_fill_zero(state.gru_ctl.data_ptr(), 4 * nctl, side.cuda_stream), "fill_zero")
        state.gru_ctl_zeroed = True
    # warm-up of the fused GLU forward (stemgnn_spectral_glu_fwd_warm): the kernel's code and block 0's weight stream into the
    # XCDs' L2 on eight CUs the GRU recurrence leaves idle; the first real launch of the step is ~10 us shorter for it
    # (exact fp32 only: the split-bf16 kernel is a third of the size and shows no first-launch penalty -- 38.1 / 37.6 us for the
    # two blocks -- while every extra launch beside the GRU forward costs that recurrence time: 219 -> 231 us with eight more)
    if batch_shape is not None and splits == 0 and os.environ.get("STEMGNN_GLU_WARM", "1") != "0":
        nwarm = lib.stemgnn_glu_warm_saved_floats(W, multi)
        if state.warm_saved is None or state.warm_saved.numel() != nwarm or state.warm_saved.device != torch.device(device):
            state.warm_saved = torch.zeros(nwarm, device=device, dtype=torch.float32)
        _lib.check(lib.stemgnn_spectral_glu_fwd_warm(packed[0].data_ptr(), split[0].data_ptr() if splits else None,
                                                     state.warm_saved.data_ptr(), batch_shape[0], batch_shape[1], W, multi,
                                                     splits, side.cuda_stream), "spectral_glu_fwd_warm")
    state.prepacked = (packed, side, blocks, (splits, split))


class SpectralHotPath(torch.autograd.Function):
    """(h, x, weight_key, weight_query, 33 params of block 0, 33 params of block 1) ->
    (sum of the two block forecasts [B,N,W], attention [N,N], mul_L [4,N,N]).

    h is the GRU output [N_seq, B, N_hid] exactly as nn.GRU returns it; x is the model input [B,W,N].
    """

    @staticmethod
    def forward(ctx, h, x, wk, wq, multi, alpha, drop_p, training, seed, state, *block_params):
        lib = _lib.load()
        state = ctx.state = _state(state)
        # `h` came from ops.GruFront of the same model (Model.hot_path says so through the state): the backward may hand
        # the GRU its output gradient in factored form instead of materialising [N,B,N]
        ctx.factored, state.gru_front_live = bool(getattr(state, "gru_front_live", False)), False
        for name, t in (("gru output", h), ("x", x), ("weight_key", wk), ("weight_query", wq)):
            _require_gpu(t, name)
        assert len(block_params) == 2 * _lib.SG_BLOCK_NPARAMS
        h = h.contiguous()
        x = x.contiguous()
        B, W, N = x.shape
        if h.shape != (N, B, N):
            raise _lib.StemGNNHipError(f"gru output must be [N,B,N]=({N},{B},{N}), got {tuple(h.shape)}")
        dev, f32 = x.device, torch.float32
        st = _stream()
        M = B * N
        blocks = [list(block_params[:33]), list(block_params[33:])]
        blocks = [[None if p is None else p.contiguous() for p in blk] for blk in blocks]
        tables = dft_tables(W, multi, dev)

        mul_L = torch.empty(4, N, N, device=dev, dtype=f32)
        attention = torch.empty(N, N, device=dev, dtype=f32)
        attn_saved = torch.empty(lib.stemgnn_attn_saved_floats(B, N), device=dev, dtype=f32)
        use_drop = bool(training) and drop_p > 0.0
        # join the side stream BEFORE the first kernel that reads `seed`: in overlap mode the dropout stream's clone /
        # increment (Model._next_seed) and the weight packing were queued there (ordering edge clone -> attention)
        pre, state.prepacked = state.prepacked, None
        if pre is not None:
            torch.cuda.current_stream().wait_stream(pre[1])
        # exact data-parallel mode: the batch mean of the attention (:140) is the one cross-sample reduction of the path;
        # averaging A | deg over the ranks between the two parts restores single-process semantics for a split batch
        exact = state.exact_group
        for part in ((3,) if exact is None else (1, 2)):
            _lib.check(lib.stemgnn_attn_laplacian_fwd(
                h.data_ptr(), wk.data_ptr(), wq.data_ptr(), float(alpha), float(drop_p), int(bool(training)),
                seed.data_ptr() if use_drop else None, B, N, attn_saved.data_ptr(), attention.data_ptr(),
                mul_L.data_ptr(), part, st), "attn_laplacian_fwd")
            if exact is not None and part == 1:
                _all_reduce_mean(attn_saved[3 * B * N:3 * B * N + N * N + N], exact)
        if _keep_attention_state:
            _last_attention_state[str(dev)] = (attn_saved[:B * N].view(B, N).clone(),
                                               attn_saved[B * N:2 * B * N].view(B, N).clone())
        if os.environ.get("STEMGNN_SPECTRAL", "cheb") == "eig":
            # north-star eigen route: L = U^T diag(lam) U, T_k = sum_e p_k(lam_e) u_e u_e^T (same function of L;
            # the backward below is the polynomial one either way -- never differentiates through eigenvectors)
            lam = torch.empty(N, device=dev, dtype=f32)
            U = torch.empty(N, N, device=dev, dtype=f32)
            escr = torch.empty(lib.stemgnn_eigh_scratch_floats(N), device=dev, dtype=f32)
            _lib.check(lib.stemgnn_eigh_fwd(mul_L.data_ptr(), lam.data_ptr(), U.data_ptr(), escr.data_ptr(), N,
                                            0, st), "eigh_fwd")
        else:
            _lib.check(lib.stemgnn_cheb_fwd(mul_L.data_ptr(), N, st), "cheb_fwd")

        fsum = torch.empty(B, N, W, device=dev, dtype=f32)
        backcast = torch.empty(B, N, W, device=dev, dtype=f32)
        packed, saved, split = [], [], []
        splits = pre[3][0] if pre is not None else glu_splits()
        n_packed = lib.stemgnn_packed_floats(W, multi)
        n_saved = lib.stemgnn_saved_floats(B, N, W, multi)
        xviews = [(x, W * N, 1, N), (backcast, N * W, W, 1)]   # X[b,n,t] strides of block 0 / block 1
        for s in range(2):
            sv = torch.empty(n_saved, device=dev, dtype=f32)
            parr = _lib.ptr_array(blocks[s])
            X, sb, sn, stt = xviews[s]
            sp = None
            if pre is not None:
                pk = pre[0][s]
                sp = pre[3][1][s] if splits else None
            else:
                pk = torch.empty(n_packed, device=dev, dtype=f32)
                _pack_block(lib, parr, tables, pk, W, multi, splits, st)
                if splits:
                    sp = _split_panels(lib, pk, W, multi, splits, dev, st)
            _lib.check(lib.stemgnn_gft_fwd(mul_L.data_ptr(), X.data_ptr(), sb, sn, stt, sv.data_ptr(), B, N, W, st),
                       "gft_fwd")
            if splits:
                _lib.check(lib.stemgnn_spectral_glu_fwd_split(pk.data_ptr(), sp.data_ptr(), sv.data_ptr(), B, N, W, multi,
                                                              splits, st), "spectral_glu_fwd_split")
            else:
                _lib.check(lib.stemgnn_spectral_glu_fwd(pk.data_ptr(), sv.data_ptr(), B, N, W, multi, st),
                           "spectral_glu_fwd")
            split.append(sp)
            _lib.check(lib.stemgnn_igft_heads_fwd(
                parr, pk.data_ptr(), sv.data_ptr(), X.data_ptr(), sb, sn, stt, fsum.data_ptr(), int(s == 1),
                backcast.data_ptr() if s == 0 else None, B, N, W, multi, st), "igft_heads_fwd")
            packed.append(pk)
            saved.append(sv)

        ctx.dims = (B, N, W, multi, float(alpha), float(drop_p), bool(training))
        ctx.blocks = blocks
        ctx.aux = (h, x, wk, wq, seed, tables, mul_L, attn_saved, backcast, packed, saved)
        ctx.split = (splits, split)
        ctx.mark_non_differentiable(attention, mul_L)
        ctx.set_materialize_grads(False)
        return fsum, attention, mul_L

    @staticmethod
    def backward(ctx, dfsum, _datt, _dmulL):
        lib = _lib.load()
        B, N, W, multi, alpha, drop_p, training = ctx.dims
        h, x, wk, wq, seed, tables, mul_L, attn_saved, backcast, packed, saved = ctx.aux
        blocks = ctx.blocks
        state = ctx.state
        splits, split = ctx.split
        dev, f32 = x.device, torch.float32
        st = _stream()
        dfsum = dfsum.contiguous()
        tail_finish, state.tail_finish = state.tail_finish, None     # FcTailMse.forward's deferred partial-sum launch
        nsplit = _NSPLIT
        n_scratch = lib.stemgnn_scratch_floats(B, N, W, multi)
        off_dG = lib.stemgnn_scratch_offset_dG(B, N, W, multi)
        n_gradpart = lib.stemgnn_gradpart_floats(W, multi, nsplit)
        dmul_L = torch.empty(4, N, N, device=dev, dtype=f32)
        dbackcast = torch.empty(B, N, W, device=dev, dtype=f32)
        xviews = [(x, W * N, 1, N), (backcast, N * W, W, 1)]
        grads = [[None] * 33, [None] * 33]
        direct_idx = set()
        for s in (1, 0):
            for i, p in enumerate(blocks[s]):
                if p is None or (s == 1 and i in (7, 8)):   # block 1's short-cut is unused (:73-74) -> grad None
                    continue
                if state.direct and p.grad is not None and p.grad.is_contiguous():
                    grads[s][i] = p.grad          # written in place by the unpack kernel
                    direct_idx.add((s, i))
                else:
                    grads[s][i] = torch.empty_like(p)
        # The weight-gradient GEMMs + un-packing feed nothing downstream in the backward pass.  When every block
        # gradient is written in place (direct-grad mode) they run on a side stream, overlapping the rest of the
        # chain -- in particular the latency-bound GRU recurrence, which leaves half of the CUs idle; whoever
        # consumes the gradients (optimizer / all-reduce, or GruFront.backward at the latest) joins the stream.
        n_block_grads = sum(g is not None for blk in grads for g in blk)
        overlap = state.overlap and len(direct_idx) == n_block_grads
        side = _side_stream(dev) if overlap else None
        main = torch.cuda.current_stream()
        keep = []
        if tail_finish is not None and not overlap:
            tail_finish(st)
            tail_finish = None
        # Scheduling of the weight-gradient work (heads + GLU weight-gradient GEMMs + un-packing), which feeds nothing
        # downstream in the backward pass.  overlap: both blocks' weight gradients go to the side stream, block 0's
        # first (it overlaps the cheb / attention chain), then block 1's, which then runs under the GRU recurrence
        # (that kernel is latency-bound on half of the CUs and reserves its CUs' LDS, so the GEMMs land on the idle
        # CUs); block 1 therefore keeps its own scratch / partial buffers until the join.
        defer_b1 = overlap
        scratch0 = torch.empty(n_scratch, device=dev, dtype=f32)
        gradpart0 = torch.empty(n_gradpart, device=dev, dtype=f32)
        bufs = {0: (scratch0, gradpart0), 1: (scratch0, gradpart0)}
        if defer_b1:
            bufs[1] = (torch.empty(n_scratch, device=dev, dtype=f32), torch.empty(n_gradpart, device=dev, dtype=f32))

        def stage_fns(s):
            scratch, gradpart = bufs[s]
            parr = _lib.ptr_array(blocks[s])
            X, sb, sn, stt = xviews[s]
            has_bc = s == 0
            garr = _lib.ptr_array(grads[s])
            keep.append((parr, garr))

            def heads(stream):                  # data part: dpF, dpB, dig, d(pre-activation) of the last GLU layer
                _lib.check(lib.stemgnn_igft_heads_bwd(
                    parr, packed[s].data_ptr(), saved[s].data_ptr(), X.data_ptr(), sb, sn, stt, dfsum.data_ptr(),
                    dbackcast.data_ptr() if has_bc else None, backcast.data_ptr() if has_bc else None,
                    scratch.data_ptr(), gradpart.data_ptr(), nsplit, 1, B, N, W, multi, stream), "igft_heads_bwd")

            def glu(stream):                    # data-gradient chain of the three GLU layers -> dG
                # (bf16x2 with 4 W multi <= 256: one fused launch on the bf16 matrix pipe, csrc/glu_fused_bf16.h -- the entry
                # point decides; measured in round 5: the per-layer split launches cost +31 us per step against the fused fp32
                # chain, the fused bf16 chain is the one that beats it)
                if splits:
                    _lib.check(lib.stemgnn_spectral_glu_dgrad_split(
                        packed[s].data_ptr(), split[s].data_ptr(), saved[s].data_ptr(), scratch.data_ptr(),
                        B, N, W, multi, splits, stream), "spectral_glu_dgrad_split")
                    return
                _lib.check(lib.stemgnn_spectral_glu_bwd(
                    packed[s].data_ptr(), saved[s].data_ptr(), scratch.data_ptr(), gradpart.data_ptr(), nsplit, 1,
                    B, N, W, multi, stream), "spectral_glu_bwd")

            def wgrad(stream, cu_percent):      # every weight gradient of the block (fused kernel + the BS head)
                # bf16x2: the fused launch's products as split-bf16 too (round 6, csrc/wgrad.h wg_stage_bf16)
                _lib.check(lib.stemgnn_block_wgrad_split(
                    parr, packed[s].data_ptr(), saved[s].data_ptr(), X.data_ptr(), sb, sn, stt, dfsum.data_ptr(),
                    int(has_bc), scratch.data_ptr(), gradpart.data_ptr(), nsplit, cu_percent, B, N, W, multi,
                    2 if splits == 2 and os.environ.get("STEMGNN_WGRAD_BF16", "1") != "0" else 0, stream), "block_wgrad")

            def unpack(stream):
                _lib.check(lib.stemgnn_block_unpack_grads(
                    gradpart.data_ptr(), nsplit, tables.data_ptr(), garr, W, multi, int(has_bc), stream),
                    "block_unpack_grads")

            return heads, glu, wgrad, unpack

        # side-stream schedule: both weight-gradient launches fork right behind block 0's data-gradient chain -- block 0's
        # runs beside the Chebyshev / attention backward chain, block 1's (sized for the CUs the GRU leaves free) under the
        # GRU recurrence.  Measured alternatives (round 3, removed in round 4): forking block 1's right behind its own chain,
        # beside block 0's MFMA-bound data-gradient kernels, +170 us per step (two GEMM streams on one chip are zero-sum);
        # forking only behind the Chebyshev backward +32 us; block 0's GFT backward ahead of the fork +9 us.  Round 5, with the
        # faster bf16 chain and weight-gradient kernels: both launches forked at the GRU backward (under the recurrence) +29 us.
        # Capture order matters inside the hipGraph step: at a fork the FIRST captured successor of a node stays on its
        # queue, every later one moves to another queue behind a cross-queue edge (~10 us).  So at every fork the main
        # stream's next kernel (the critical chain) is queued before the side stream's work that forks at the same node.
        # Both blocks' shares of d(mul_L): round 6 -- ONE product behind block 0's data-gradient chain (stemgnn_gft_bwd_dt2, K =
        # the two blocks' (b, t) ranges back to back) instead of block 1's product on the side branch + block 0's accumulating
        # product on the chain: the fork behind block 1's dX product and the join ahead of block 0's product were ~15 us of
        # cross-queue latency on the critical chain.
        dt1 = None
        for s in (1, 0):
            scratch = bufs[s][0]
            dG = scratch[off_dG:]
            X, sb, sn, stt = xviews[s]
            heads, glu, wgrad, unpack = stage_fns(s)
            heads(st)
            glu(st)
            if not overlap:
                wgrad(st, 100)
                unpack(st)
                _lib.check(lib.stemgnn_gft_bwd(
                    mul_L.data_ptr(), X.data_ptr(), sb, sn, stt, dG.data_ptr(),
                    dbackcast.data_ptr() if s == 1 else None, dmul_L.data_ptr(), int(s == 0), B, N, W, st), "gft_bwd")
            elif s == 1:
                # block 1: only its data gradient (-> dbackcast) feeds block 0's backward
                _lib.check(lib.stemgnn_gft_bwd(mul_L.data_ptr(), X.data_ptr(), sb, sn, stt, dG.data_ptr(),
                                               dbackcast.data_ptr(), None, 0, B, N, W, st), "gft_bwd dX")
                dt1 = (X, sb, sn, stt, dG)
            else:
                fork = torch.cuda.Event()                # every data-gradient chain is queued
                fork.record(main)
                X1, sb1, sn1, stt1, dG1 = dt1
                _lib.check(lib.stemgnn_gft_bwd_dt2(X.data_ptr(), sb, sn, stt, dG.data_ptr(), X1.data_ptr(), sb1, sn1, stt1,
                                                   dG1.data_ptr(), dmul_L.data_ptr(), B, N, W, st), "gft_bwd_dt2")
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    sst = side.cuda_stream
                    for ss in (0, 1):
                        _h, _g, w2, u2 = (heads, glu, wgrad, unpack) if ss == 0 else stage_fns(1)
                        w2(sst, _wg_cu(ss, B, N))
                        u2(sst)
                        if state.side_probe is not None:
                            state.side_probe.append(lambda stream, w2=w2, pc=_wg_cu(ss, B, N): w2(stream, pc))
                            state.side_probe.append(u2)
                    if tail_finish is not None:          # loss + fc gradients: ahead of the hook (fc is part of its range)
                        tail_finish(sst)
                        keep.append(tail_finish)         # its buffers stay alive until the join
                        if state.side_probe is not None:
                            state.side_probe.append(tail_finish)
                    if state.block_grads_hook is not None:
                        state.block_grads_hook()         # data-parallel: reduce the finished range under the GRU recurrence
                keep.append(bufs)                        # alive until the join
        if overlap:
            state.pending = (side, (keep, packed, saved, split, backcast, dfsum, dbackcast))
        dL = torch.empty(N, N, device=dev, dtype=f32)
        cheb_scratch = torch.empty(2 * N * N, device=dev, dtype=f32)
        _lib.check(lib.stemgnn_cheb_bwd(mul_L.data_ptr(), dmul_L.data_ptr(), dL.data_ptr(), cheb_scratch.data_ptr(),
                                        N, st), "cheb_bwd")
        # factored: dh[s,b,i] = dkey[b,i] wk[s] + dquery[b,i] wq[s] goes to the GRU backward as its two [B,N] factors
        # (stemgnn_gru_bwd_rank2); the kernel that would materialise the 6.6 MB tensor (and form dwk / dwq on the way) leaves
        # the critical chain, dwk / dwq come from a small kernel of their own on the side stream
        factored = ctx.factored and ctx.needs_input_grad[0] and bool(lib.stemgnn_gru_bwd_rank2_ok(B, N))
        dh = torch.empty(1, device=dev, dtype=f32).expand(N, B, N) if factored else torch.empty_like(h)   # placeholder: never read
        kq_direct = state.direct and wk.grad is not None and wq.grad is not None
        dwk = wk.grad if kq_direct else torch.empty_like(wk)
        dwq = wq.grad if kq_direct else torch.empty_like(wq)
        attn_scratch = torch.empty(lib.stemgnn_attn_scratch_floats(B, N, _NCHUNK), device=dev, dtype=f32)
        use_drop = training and drop_p > 0.0
        # exact mode: d(loss_global)/dA = mean over ranks of the local dA, and the flat gradient bucket is AVERAGED later,
        # so every rank back-propagates the rank-mean of dA / B through its own samples
        exact = state.exact_group
        # round 6: with the factored form on the side-stream schedule the chunk sum of dquery is not a launch of its own on the
        # chain -- the GRU backward's zero-fill launch carries it (stemgnn_gru_bwd_rank2_dq), the key / query weight gradients
        # on the side branch sum their own copy (same fixed order, same bits)
        follower = bool(overlap and ctx.factored and state.gru_ctl_zeroed and lib.stemgnn_gru_bwd_overlap_ok(B, N, N, W))
        dq_parts = bool(factored and overlap and kq_direct and not follower)      # (the follower's _begin call takes dquery reduced)
        for part in ((3,) if exact is None else (1, 2)):
            _lib.check(lib.stemgnn_attn_laplacian_bwd(
                dL.data_ptr(), h.data_ptr(), wk.data_ptr(), wq.data_ptr(), alpha, drop_p, int(training),
                seed.data_ptr() if use_drop else None, B, N, attn_saved.data_ptr(), attn_scratch.data_ptr(), _NCHUNK,
                None if factored else dh.data_ptr(), dwk.data_ptr(), dwq.data_ptr(),
                part | ((4 | (8 if dq_parts else 0)) if factored and part != 1 else 0), st), "attn_laplacian_bwd")
            if exact is not None and part == 1:
                _all_reduce_mean(attn_scratch[:N * N], exact)
        if factored:
            after = None
            if overlap and kq_direct:       # behind block 1's weight gradients on the side stream, under the GRU recurrence;
                kq_ready = torch.cuda.Event()                         # queued by GruFront.backward BEHIND its own launches
                kq_ready.record(main)                                 # (capture order, see above)
                pend, hh, kdwk, kdwq = state.pending, h, dwk, dwq
                dq2 = torch.empty(B, N, device=dev, dtype=f32) if dq_parts else None

                def kq_wgrad(stream):
                    if dq2 is not None:
                        _lib.check(lib.stemgnn_attn_dquery_reduce(attn_scratch.data_ptr(), B, N, _NCHUNK, dq2.data_ptr(), stream),
                                   "attn_dquery_reduce")
                        _lib.check(lib.stemgnn_keyquery_wgrad2(hh.data_ptr(), attn_scratch.data_ptr() + 4 * N * N, dq2.data_ptr(),
                                                               kdwk.data_ptr(), kdwq.data_ptr(), B, N, stream), "keyquery_wgrad2")
                        return
                    _lib.check(lib.stemgnn_keyquery_wgrad(hh.data_ptr(), attn_scratch.data_ptr(), kdwk.data_ptr(),
                                                          kdwq.data_ptr(), B, N, stream), "keyquery_wgrad")

                def after():
                    side.wait_event(kq_ready)
                    with torch.cuda.stream(side):
                        kq_wgrad(side.cuda_stream)
                    pend[1][0].append((attn_scratch, hh, dq2))
                if state.side_probe is not None:
                    state.side_probe.append(kq_wgrad)
            state.dh_factors = (attn_scratch, B, N, wk, wq, after, _NCHUNK if dq_parts else 0)
            if after is None:
                _lib.check(lib.stemgnn_keyquery_wgrad(h.data_ptr(), attn_scratch.data_ptr(), dwk.data_ptr(), dwq.data_ptr(),
                                                      B, N, st), "keyquery_wgrad")
        for s_, i_ in direct_idx:
            grads[s_][i_] = None                  # already in p.grad: nothing for autograd to accumulate
        if kq_direct:
            dwk = dwq = None
        return (dh, None, dwk, dwq, None, None, None, None, None, None, *grads[0], *grads[1])


# ---------------------------------------------------------------------------------------------------------------
# data path either side of the hot path: thin wrappers over csrc/data.hip

class MSELossFn(torch.autograd.Function):
    """nn.MSELoss(reduction='mean') of the driver (reference models/handler.py:140,162) as two fixed-order kernels."""

    @staticmethod
    def forward(ctx, forecast, target, loss_out=None, accum=None):
        """loss_out: optional float32 scalar to write the loss into (a static buffer); accum: optional float64 device
        scalar that receives `+= loss` inside the reduction kernel (epoch loss sum without extra launches)."""
        lib = _lib.load()
        _require_gpu(forecast, "forecast")
        _require_gpu(target, "target")
        if forecast.shape != target.shape:
            raise _lib.StemGNNHipError(f"MSE: shapes differ {tuple(forecast.shape)} vs {tuple(target.shape)}")
        forecast, target = forecast.contiguous(), target.contiguous()
        scratch = torch.empty(lib.stemgnn_mse_scratch_floats(), device=forecast.device, dtype=torch.float32)
        loss = loss_out if loss_out is not None else torch.empty((), device=forecast.device, dtype=torch.float32)
        if accum is not None and (accum.dtype != torch.float64 or accum.device != forecast.device):
            raise _lib.StemGNNHipError("MSE: accum must be a float64 scalar on the forecast's device")
        _lib.check(lib.stemgnn_mse_fwd(forecast.data_ptr(), target.data_ptr(), forecast.numel(), scratch.data_ptr(),
                                       loss.data_ptr(), accum.data_ptr() if accum is not None else None, _stream()),
                   "mse_fwd")
        ctx.save_for_backward(forecast, target)
        ctx.set_materialize_grads(False)
        if loss_out is not None:
            ctx.mark_dirty(loss_out)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        forecast, target = ctx.saved_tensors
        grad_loss = grad_loss.contiguous()
        dforecast = torch.empty_like(forecast)
        _lib.check(lib.stemgnn_mse_bwd(forecast.data_ptr(), target.data_ptr(), forecast.numel(), grad_loss.data_ptr(),
                                       dforecast.data_ptr(), _stream()), "mse_bwd")
        return dforecast, None, None, None


def mse_loss(forecast, target, loss_out=None, accum=None):
    return MSELossFn.apply(forecast, target, loss_out, accum)


class MSELoss(torch.nn.Module):
    """Drop-in for ``nn.MSELoss(reduction='mean')`` at models/handler.py:140."""

    def __init__(self, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise ValueError("stemgnn_amd.ops.MSELoss implements reduction='mean' only (the reference's setting)")

    def forward(self, forecast, target):
        return MSELossFn.apply(forecast, target)


def normalize_series(raw, sub, div, clip01):
    """raw [T,N] float64 (device) -> float32 [T,N]: (raw - sub[n]) / div[n] in fp64, optional clip to [0,1]."""
    lib = _lib.load()
    if not raw.is_cuda or raw.dtype != torch.float64:
        raise _lib.StemGNNHipError("normalize_series: raw must be a float64 tensor on a HIP device")
    raw, sub, div = raw.contiguous(), sub.contiguous(), div.contiguous()
    T, N = raw.shape
    out = torch.empty(T, N, device=raw.device, dtype=torch.float32)
    _lib.check(lib.stemgnn_normalize_series(raw.data_ptr(), sub.data_ptr(), div.data_ptr(), int(bool(clip01)),
                                            out.data_ptr(), T, N, _stream()), "normalize_series")
    return out


_gather_status = {}


def window_gather(series, hi, W, H, x=None, y=None):
    """series [T,N] fp32 resident, hi [B] int64 (device): x[b] = series[hi-W:hi], y[b] = series[hi:hi+H]."""
    lib = _lib.load()
    _require_gpu(series, "series")
    if hi.dtype != torch.int64 or hi.device != series.device:
        raise _lib.StemGNNHipError("window_gather: hi must be an int64 tensor on the series' device")
    T, N = series.shape
    B = hi.numel()
    if x is None:
        x = torch.empty(B, W, N, device=series.device, dtype=torch.float32)
    if y is None:
        y = torch.empty(B, H, N, device=series.device, dtype=torch.float32)
    key = str(series.device)
    st = _gather_status.get(key)
    if st is None:
        st = _gather_status[key] = torch.zeros(1, dtype=torch.int32, device=series.device)
    _lib.check(lib.stemgnn_window_gather(series.data_ptr(), hi.data_ptr(), x.data_ptr(), y.data_ptr(), B, W, H, N, T,
                                         st.data_ptr(), _stream()), "window_gather")
    return x, y


def window_gather_queue(series, order, queue, B, W, H, x, y):
    """Iterator form (stemgnn_window_gather_queue): the next B windows of `order` (int64, device) at the device-side
    position queue[0]; the kernel advances the position itself.  queue: int64[4] = {position, 0, count, -}."""
    lib = _lib.load()
    _require_gpu(series, "series")
    for t, n in ((order, "order"), (queue, "queue")):
        if t.dtype != torch.int64 or t.device != series.device or not t.is_contiguous():
            raise _lib.StemGNNHipError(f"window_gather_queue: {n} must be a contiguous int64 tensor on the series' device")
    T, N = series.shape
    key = str(series.device)
    st = _gather_status.get(key)
    if st is None:
        st = _gather_status[key] = torch.zeros(1, dtype=torch.int32, device=series.device)
    _lib.check(lib.stemgnn_window_gather_queue(series.data_ptr(), order.data_ptr(), queue.data_ptr(), x.data_ptr(),
                                               y.data_ptr(), B, W, H, N, T, st.data_ptr(), _stream()), "window_gather_queue")
    return x, y


def check_gather_status(device):
    """Raise if any window_gather since the last check saw an out-of-range index (one sync; call per epoch)."""
    st = _gather_status.get(str(device))
    if st is not None:
        v = int(st.item())
        if v != 0:
            st.zero_()
            raise IndexError("window_gather: window index outside the series" if v & 1 else
                             "window_gather_queue: stepped past the end of the loaded order")


def roll_window(inputs, forecast, forecast_steps, step, horizon):
    """One iteration of the reference's rolling inference (models/handler.py:56-61); returns the next inputs."""
    lib = _lib.load()
    _require_gpu(inputs, "inputs")
    _require_gpu(forecast, "forecast")
    B, W, N = inputs.shape
    L = forecast.shape[1]
    if L == 0:
        raise Exception("Get blank inference result")                    # handler.py:54-55
    inputs, forecast = inputs.contiguous(), forecast.contiguous()
    nxt = torch.empty_like(inputs)
    _lib.check(lib.stemgnn_roll_window(inputs.data_ptr(), forecast.data_ptr(), nxt.data_ptr(),
                                       forecast_steps.data_ptr(), B, W, L, N, int(step), int(horizon), _stream()),
               "roll_window")
    return nxt


def eval_metrics(target, forecast, mul=None, add=None):
    """target / forecast [count,H,N] fp32 -> float64 device vector
    overall[3] | by_node[3][N] | by_step[3][H] | by_step_node[3][H][N]  (MAPE, MAE, RMSE each)."""
    lib = _lib.load()
    _require_gpu(target, "target")
    _require_gpu(forecast, "forecast")
    if target.shape != forecast.shape or target.dim() != 3:
        raise _lib.StemGNNHipError("eval_metrics: target/forecast must both be [count, time_step, node]")
    target, forecast = target.contiguous(), forecast.contiguous()
    C, H, N = target.shape
    dev = target.device
    scratch = torch.empty(lib.stemgnn_eval_scratch_doubles(C, H, N), device=dev, dtype=torch.float64)
    out = torch.empty(lib.stemgnn_eval_out_doubles(H, N), device=dev, dtype=torch.float64)
    if mul is not None:
        mul = mul.to(device=dev, dtype=torch.float64).contiguous()
        add = add.to(device=dev, dtype=torch.float64).contiguous()
    _lib.check(lib.stemgnn_eval_metrics(target.data_ptr(), forecast.data_ptr(),
                                        mul.data_ptr() if mul is not None else None,
                                        add.data_ptr() if add is not None else None,
                                        C, H, N, scratch.data_ptr(), out.data_ptr(), _stream()), "eval_metrics")
    return out
