"""CPU: pin the oracle restatement against outputs of the REAL reference (tests/golden/*.npz),
and -- when /root/reference is mounted -- against the live reference on fresh inputs."""
import numpy as np
import pytest
import torch

from oracle import stemgnn_oracle as O
from oracle.ref_shim import load_reference_model_module, reference_available
from tests.util import golden_cases, hash_seed, load_golden, relerr

TOL32 = 2e-5   # oracle fp32 vs reference fp32 (same op order up to FFT/GRU library details)
TOL64 = 2e-5   # oracle fp64 vs reference fp32: the reference's own fp32 round-off


def _run_oracle(name, dtype):
    z, cfg = load_golden(name)
    sd = O.det_state_dict(cfg["N"], cfg["W"], cfg["multi"], cfg["H"], seed=hash_seed(name), dtype=dtype)
    x = torch.from_numpy(z["x"]).to(dtype)
    y = torch.from_numpy(z["y"]).to(dtype)
    kw = {}
    if cfg["mode"] == "mask":
        kw = dict(drop_mask=torch.from_numpy(z["drop_mask"]).to(dtype), drop_p=0.5)
    loss, forecast, att, grads = O.loss_and_grads(x, y, sd, **kw)
    return z, cfg, loss, forecast, att, grads


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.float64, TOL64)])
def test_oracle_matches_reference_golden(name, dtype, tol):
    z, cfg, loss, forecast, att, grads = _run_oracle(name, dtype)
    assert forecast.shape == z["forecast"].shape
    assert relerr(forecast, z["forecast"]) < tol
    assert relerr(att, z["attention"]) < tol
    assert abs(float(loss) - float(z["loss"])) < tol * max(1.0, abs(float(z["loss"])))
    for k, g in grads.items():
        if "gradnone." + k in z:
            # the reference leaves these without a gradient (block 1's unused backcast_short_cut, :73-74)
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert g is not None, k
        if "grad." + k in z:
            assert relerr(g, z["grad." + k]) < 5 * tol, k
        else:
            head = z["gradhead." + k]
            nrm, sm, mx = z["gradstat." + k]
            assert np.abs(g.reshape(-1)[:64].double().numpy() - head).max() <= 5 * tol * mx, k
            assert abs(float(g.double().pow(2).sum().sqrt()) - nrm) <= 5 * tol * nrm, k


@pytest.mark.parametrize("name", golden_cases())
def test_mul_L_golden(name):
    z, cfg = load_golden(name)
    sd = O.det_state_dict(cfg["N"], cfg["W"], cfg["multi"], cfg["H"], seed=hash_seed(name))
    x = torch.from_numpy(z["x"])
    kw = {}
    if cfg["mode"] == "mask":
        kw = dict(drop_mask=torch.from_numpy(z["drop_mask"]), drop_p=0.5)
    _, _, mul_L = O.hot_path(O.gru_front(x, sd), x, sd, **kw)
    assert relerr(mul_L, z["mul_L"]) < TOL32
    assert float(mul_L[0].abs().max()) == 0.0          # T0 is zeros, not I (models/base_model.py:129)
    # eigen route reproduces the same polynomial of L (north-star eigensolver identity)
    _, _, mul_L_eig = O.hot_path(O.gru_front(x, sd), x, sd, spectral="eig", **kw)
    assert relerr(mul_L_eig, z["mul_L"]) < 1e-4


def test_dead_rows_have_zero_grad():
    """C2R ignores bins > n/2 and Im of DC/Nyquist -> dead output rows of the last GLUs."""
    name = "small_train_p0"
    z, cfg, *_, grads = _run_oracle(name, torch.float64)
    Wm = cfg["W"] * cfg["multi"]
    g_re = grads["stock_block.0.GLUs.4.linear_left.weight"].reshape(4, Wm, -1)
    g_im = grads["stock_block.0.GLUs.5.linear_left.weight"].reshape(4, Wm, -1)
    assert float(g_re[:, Wm // 2 + 1:].abs().max()) == 0.0
    assert float(g_re[:, : Wm // 2 + 1].abs().max()) > 0.0
    assert float(g_im[:, Wm // 2:].abs().max()) == 0.0 and float(g_im[:, 0].abs().max()) == 0.0


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted (GPU box)")
@pytest.mark.parametrize("N,W,multi,H,B", [(11, 12, 5, 3, 3), (16, 8, 2, 1, 6)])
def test_oracle_matches_live_reference(N, W, multi, H, B):
    ref = load_reference_model_module()
    torch.manual_seed(123)
    model = ref.Model(N, 2, W, multi, horizon=H, dropout_rate=0.0)   # reference's own random init
    model.train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert list(sd.keys()) == list(O.param_shapes(N, W, multi, H).keys())
    assert all(tuple(sd[k].shape) == s for k, s in O.param_shapes(N, W, multi, H).items())
    x, y = torch.randn(B, W, N), torch.randn(B, H, N)
    forecast, att = model(x)
    loss = torch.nn.functional.mse_loss(forecast, y)
    loss.backward()
    o_loss, o_forecast, o_att, o_grads = O.loss_and_grads(x, y, sd)
    assert relerr(o_forecast, forecast.detach()) < TOL32
    assert relerr(o_att, att.detach()) < TOL32
    for k, p in model.named_parameters():
        if p.grad is None:
            assert o_grads[k] is None or float(o_grads[k].abs().max()) == 0.0
        else:
            assert relerr(o_grads[k], p.grad) < 5 * TOL32, k


def test_written_out_gru_cell_matches_aten_gru():
    """oracle.gru_manual (the device-agnostic written-out cell the LARGE GPU parity cases evaluate in fp64 on the
    device) against ATen's _VF.gru -- what the reference's nn.GRU runs (models/base_model.py:92,137): output and all
    four parameter gradients, fp64."""
    torch.manual_seed(5)
    S, B, W, H = 9, 4, 5, 7
    gru = torch.nn.GRU(W, H).double()
    seq = torch.randn(S, B, W, dtype=torch.float64)
    dh = torch.randn(S, B, H, dtype=torch.float64)
    out, _ = gru(seq)
    out.backward(dh)
    ps = [p.detach().clone().requires_grad_(True) for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
    mine = O.gru_manual(seq, *ps)
    mine.backward(dh)
    assert relerr(mine.detach(), out.detach()) < 1e-13
    for a, b in zip(ps, (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)):
        assert relerr(a.grad, b.grad) < 1e-12
