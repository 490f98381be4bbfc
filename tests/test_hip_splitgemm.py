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


# ---- the split arithmetic inside the model (STEMGNN_DTYPE, csrc/gemm2s.h) ---------------------------------------------
# (N, W, multi, H, B): PEMS07 shape (the bench workload), ECG shape, ragged rows, odd W*multi (falls back to fp32 kernels)
SPLIT_CASES = [(228, 12, 5, 3, 32), (140, 12, 5, 3, 32), (33, 12, 5, 1, 5), (19, 5, 3, 2, 3)]


@pytest.mark.parametrize("dtype,tol", [("bf16x3", 1e-4), ("bf16x2", 1e-4)])
@pytest.mark.parametrize("N,W,multi,H,B", SPLIT_CASES)
def test_model_with_split_bf16_glu_matches_oracle(monkeypatch, dtype, tol, N, W, multi, H, B):
    """Same comparison as test_hip_parity.test_oracle_parity_fwd_bwd with the GLU forward / data-gradient layers on the
    split-bf16 kernel: bf16x3 is fp32 class; bf16x2 (2^-16 per product) must still meet north_star's 1e-4."""
    from oracle import stemgnn_oracle as O
    from stemgnn_amd import Model

    monkeypatch.setenv("STEMGNN_DTYPE", dtype)
    sd = O.det_state_dict(N, W, multi, H, seed=N + B)
    torch.manual_seed(N * 7 + B)
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)
    model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0)
    model.load_state_dict(sd)
    model.to("cuda:0").train()
    forecast, att = model(x.cuda())
    torch.nn.functional.mse_loss(forecast, y.cuda()).backward()
    torch.cuda.synchronize()
    o_loss, o_forecast, o_att, o_grads = O.loss_and_grads(x, y, sd)
    errs = {"forecast": relerr(forecast, o_forecast), "attention": relerr(att, o_att)}
    for k, p in model.named_parameters():
        if o_grads[k] is not None:
            errs["grad." + k] = relerr(p.grad, o_grads[k])
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"{dtype} N={N} W={W} multi={multi} B={B}: worst norm-relative error {worst[1]:.2e} ({worst[0]})")
    assert worst[1] < tol, errs


def test_split_dtype_graph_step_trains_like_fp32(monkeypatch):
    """The hipGraph train step under STEMGNN_DTYPE=bf16x3 (weight split on the side stream beside the packing): ten
    steps from the same seed track the fp32 step's losses."""
    from oracle import stemgnn_oracle as O
    from stemgnn_amd import Model
    from stemgnn_amd.engine import TrainStep
    from stemgnn_amd.optim import FusedRMSprop

    N, W, multi, H, B = 60, 12, 5, 3, 16
    losses = {}
    for dtype in ("f32", "bf16x3"):
        monkeypatch.setenv("STEMGNN_DTYPE", dtype)
        sd = O.det_state_dict(N, W, multi, H, seed=3)
        model = Model(N, 2, W, multi, horizon=H, dropout_rate=0.0)
        model.load_state_dict(sd)
        model.to("cuda:0").train()
        opt = FusedRMSprop(model.parameters(), lr=1e-3, alpha=0.99, eps=1e-8)
        torch.manual_seed(11)
        series = torch.randn(400, N).cuda()
        step = TrainStep(model, opt, B, W, H, N, series=series)
        g = torch.Generator().manual_seed(5)
        out = []
        for _ in range(10):
            hi = (torch.randint(W, 400 - H, (B,), generator=g)).cuda()
            step.run_indices(hi)
            out.append(float(step.loss.item()))
        assert step.mode.startswith("hipgraph"), step.mode
        losses[dtype] = out
    for a, b in zip(losses["f32"], losses["bf16x3"]):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (losses["f32"], losses["bf16x3"])


@pytest.mark.parametrize("N,W,multi,B", [(228, 12, 5, 32), (140, 12, 5, 7), (33, 12, 5, 5), (50, 8, 2, 9), (19, 5, 3, 3), (64, 16, 4, 4)])
def test_fused_bf16_forward_stage_matches_the_fp32_fused_forward(monkeypatch, N, W, multi, B):
    """csrc/glu_fused_bf16.h (round 5): the three GLU layers of a block in ONE launch with split-bf16 products inside, through
    the C ABI on random panels, against the exact-fp32 fused kernel on the same buffers: every saved `out` / `gate` tensor
    within 1e-4 norm-relative (2^-16 per product, three layers deep); and the per-layer split launches
    (STEMGNN_GLU_FUSED=0) land in the same place.  Shapes: both channel-group counts (4 W multi = 240 / 64 / 60 / 256),
    ragged last row blocks, K = 3 W not a multiple of 16."""
    from stemgnn_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    if not lib.stemgnn_glu_fused_bf16_ok(W, multi, 2):
        pytest.skip("fused bf16 forward does not apply to this shape")
    g = torch.Generator().manual_seed(N * 3 + W)
    packed = (torch.randn(lib.stemgnn_packed_floats(W, multi), generator=g) * 0.08).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.stemgnn_glu_fused_repack(packed.data_ptr(), W, multi, st), "repack")
    split = torch.empty(lib.stemgnn_glu_split_floats(W, multi, 2), device=dev)
    _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st), "split_panels")
    base = torch.randn(lib.stemgnn_saved_floats(B, N, W, multi), generator=g).to(dev)
    ref, got, per_layer = base.clone(), base.clone(), base.clone()
    _lib.check(lib.stemgnn_spectral_glu_fwd(packed.data_ptr(), ref.data_ptr(), B, N, W, multi, st), "fwd fp32")
    _lib.check(lib.stemgnn_spectral_glu_fwd_split(packed.data_ptr(), split.data_ptr(), got.data_ptr(), B, N, W, multi, 2, st),
               "fwd fused bf16")
    monkeypatch.setenv("STEMGNN_GLU_FUSED", "0")           # (the plane sets of the per-layer kernels are only written in this mode)
    _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st), "split_panels")
    _lib.check(lib.stemgnn_spectral_glu_fwd_split(packed.data_ptr(), split.data_ptr(), per_layer.data_ptr(), B, N, W, multi, 2,
                                                  st), "fwd per-layer bf16")
    torch.cuda.synchronize()
    assert not torch.equal(ref, base)                       # something was written
    assert torch.isfinite(got).all()
    # per saved tensor: G (untouched), then out / gate per branch and layer, then ig / fs (untouched)
    M, CP, KG = B * N, (4 * W * multi + 15) // 16 * 16, 3 * W
    off, worst = M * KG, 0.0
    assert torch.equal(got[:off], base[:off])
    while off < ref.numel():
        n = min(M * 16, ref.numel() - off)                  # 16-column strips: a local error cannot hide in a big tensor's norm
        seg_ref = ref[off:off + n]
        if float(seg_ref.abs().max()) > 0:
            worst = max(worst, relerr(got[off:off + n], seg_ref), )
            assert relerr(per_layer[off:off + n], seg_ref) < 2e-4
        off += n
    print(f"N={N} W={W} multi={multi} B={B} (CP={CP}): fused bf16x2 forward vs fp32, worst strip {worst:.2e}")
    assert worst < 1e-4


@pytest.mark.parametrize("N,W,multi,B", [(228, 12, 5, 32), (140, 12, 5, 7), (33, 12, 5, 5), (50, 8, 2, 9), (19, 5, 3, 3), (64, 16, 4, 4)])
def test_fused_bf16_data_gradient_chain_matches_the_fp32_chain(N, W, multi, B):
    """csrc/glu_fused_bf16.h, sg_glu_fused_dgrad_bf16_kernel (round 5): d(pre-activation) of layer 2 -> 1 -> 0 -> dG in ONE
    launch with split-bf16 products inside, through the C ABI on random buffers, against the fused fp32 chain on the same
    buffers: d(pre-activation) of layers 1 / 0 and both dG slabs within 1e-4 norm-relative, compared in strips."""
    from stemgnn_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    if not lib.stemgnn_glu_fused_bf16_ok(W, multi, 2):
        pytest.skip("fused bf16 kernels do not apply to this shape")
    g = torch.Generator().manual_seed(N * 5 + W)
    packed = (torch.randn(lib.stemgnn_packed_floats(W, multi), generator=g) * 0.08).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.stemgnn_glu_fused_repack(packed.data_ptr(), W, multi, st), "repack")
    split = torch.empty(lib.stemgnn_glu_split_floats(W, multi, 2), device=dev)
    _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st), "split_panels")
    saved = torch.rand(lib.stemgnn_saved_floats(B, N, W, multi), generator=g).to(dev)        # out / gate in (0, 1)
    base = (torch.randn(lib.stemgnn_scratch_floats(B, N, W, multi), generator=g) * 0.1).to(dev)
    ref, got = base.clone(), base.clone()
    gradpart = torch.empty(lib.stemgnn_gradpart_floats(W, multi, ops._NSPLIT), device=dev)
    _lib.check(lib.stemgnn_spectral_glu_bwd(packed.data_ptr(), saved.data_ptr(), ref.data_ptr(), gradpart.data_ptr(), ops._NSPLIT, 1,
                                            B, N, W, multi, st), "fp32 chain")
    _lib.check(lib.stemgnn_spectral_glu_dgrad_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), got.data_ptr(), B, N, W,
                                                    multi, 2, st), "bf16 chain")
    torch.cuda.synchronize()
    assert not torch.equal(ref, base) and torch.isfinite(got).all()
    M, worst, off = B * N, 0.0, 0
    while off < ref.numel():
        n = min(M * 16, ref.numel() - off)
        seg = ref[off:off + n]
        if float(seg.abs().max()) > 0:
            worst = max(worst, relerr(got[off:off + n], seg))
        off += n
    print(f"N={N} W={W} multi={multi} B={B}: fused bf16x2 data-gradient chain vs fp32, worst strip {worst:.2e}")
    assert worst < 1e-4


def test_per_layer_split_entries_refuse_a_fused_only_buffer(monkeypatch):
    """stemgnn_glu_split_panels skips the per-layer plane sets where both fused bf16 forms apply; if
    STEMGNN_GLU_FUSED is then flipped before the use (it is read per call) or the chain's scratch is not 16-byte aligned,
    the per-layer kernels would read planes nobody wrote.  They refuse (SG_EINVAL) instead; after a pack in per-layer mode
    the same calls run."""
    from stemgnn_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    N, W, multi, B = 33, 12, 5, 5
    assert lib.stemgnn_glu_fused_bf16_ok(W, multi, 2)
    g = torch.Generator().manual_seed(5)
    packed = (torch.randn(lib.stemgnn_packed_floats(W, multi), generator=g) * 0.08).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    split = torch.empty(lib.stemgnn_glu_split_floats(W, multi, 2), device=dev)
    saved = torch.rand(lib.stemgnn_saved_floats(B, N, W, multi), generator=g).to(dev)
    scratch = (torch.randn(lib.stemgnn_scratch_floats(B, N, W, multi) + 4, generator=g) * 0.1).to(dev)
    monkeypatch.delenv("STEMGNN_GLU_FUSED", raising=False)
    _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st), "split_panels")       # fused-only pack
    # (a) misaligned scratch: the fused chain cannot take it, the per-layer kernels have no planes
    assert lib.stemgnn_spectral_glu_dgrad_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), scratch.data_ptr() + 4,
                                                B, N, W, multi, 2, st) == _lib.SG_EINVAL
    # (b) the switch flipped between pack and use
    monkeypatch.setenv("STEMGNN_GLU_FUSED", "0")
    assert lib.stemgnn_spectral_glu_dgrad_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                                B, N, W, multi, 2, st) == _lib.SG_EINVAL
    assert lib.stemgnn_spectral_glu_fwd_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), B, N, W, multi, 2,
                                              st) == _lib.SG_EINVAL
    # packed again in per-layer mode: the planes exist, the same calls run
    _lib.check(lib.stemgnn_glu_split_panels(packed.data_ptr(), split.data_ptr(), W, multi, 2, st), "split_panels")
    _lib.check(lib.stemgnn_spectral_glu_fwd_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), B, N, W, multi, 2, st),
               "per-layer forward")
    _lib.check(lib.stemgnn_spectral_glu_dgrad_split(packed.data_ptr(), split.data_ptr(), saved.data_ptr(), scratch.data_ptr(),
                                                    B, N, W, multi, 2, st), "per-layer chain")
    torch.cuda.synchronize()
    assert torch.isfinite(saved).all()


@pytest.mark.parametrize("N,W,multi,B", [(228, 12, 5, 32), (140, 12, 5, 7), (33, 12, 5, 5), (50, 8, 2, 9), (64, 16, 4, 4)])
def test_split_bf16_weight_gradients_match_the_fp32_launch(N, W, multi, B):
    """csrc/wgrad.h wg_stage_bf16 (round 6): the fused weight-gradient launch of a block -- six GLU products and the heads'
    products -- as three-term split-bf16 on v_mfma_f32_32x32x16_bf16, through the C ABI on random saved / scratch buffers,
    against the exact-fp32 launch on the same buffers: every parameter gradient (after stemgnn_block_unpack_grads) within 1e-4
    norm-relative; two launches of the split form agree bit for bit (fixed-order split reduction); both tile shapes (the
    256 x 64 one for the layer-0 products), ragged K ranges (M = B N not a multiple of 16) and the ones column (bias)."""
    from stemgnn_amd import StockBlockLayer, _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(N + W)
    blk = StockBlockLayer(W, N, multi, stack_cnt=0).to(dev)
    params = blk.hip_params()
    parr = _lib.ptr_array(params)
    tables = ops.dft_tables(W, multi, dev)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(N * 7 + W)
    packed = torch.empty(lib.stemgnn_packed_floats(W, multi), device=dev)
    _lib.check(lib.stemgnn_block_pack(parr, tables.data_ptr(), packed.data_ptr(), W, multi, st), "pack")
    saved = torch.rand(lib.stemgnn_saved_floats(B, N, W, multi), generator=g).to(dev)
    scratch = (torch.randn(lib.stemgnn_scratch_floats(B, N, W, multi), generator=g) * 0.1).to(dev)
    X = torch.randn(B, N, W, generator=g).to(dev)
    dfo = (torch.randn(B, N, W, generator=g) * 0.1).to(dev)
    ns = ops._NSPLIT

    def run(splits):
        gradpart = torch.zeros(lib.stemgnn_gradpart_floats(W, multi, ns), device=dev)
        _lib.check(lib.stemgnn_block_wgrad_split(parr, packed.data_ptr(), saved.data_ptr(), X.data_ptr(), N * W, W, 1,
                                                 dfo.data_ptr(), 1, scratch.data_ptr(), gradpart.data_ptr(), ns, 100, B, N, W,
                                                 multi, splits, st), "block_wgrad_split")
        grads = [None if p is None else torch.zeros_like(p) for p in params]
        _lib.check(lib.stemgnn_block_unpack_grads(gradpart.data_ptr(), ns, tables.data_ptr(), _lib.ptr_array(grads), W, multi, 1,
                                                  st), "unpack")
        torch.cuda.synchronize()
        return grads
    ref, got, again = run(0), run(2), run(2)
    worst = 0.0
    for i, (a, b, c) in enumerate(zip(ref, got, again)):
        if a is None:
            continue
        assert torch.isfinite(b).all(), i
        assert torch.equal(b, c), i                         # launch-to-launch determinism of the split form
        if float(a.abs().max()) > 0:
            worst = max(worst, relerr(b, a))
    print(f"N={N} W={W} multi={multi} B={B}: split-bf16 weight gradients vs fp32, worst parameter {worst:.2e}")
    assert 0 < worst < 1e-4
