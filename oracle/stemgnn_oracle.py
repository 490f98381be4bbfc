"""TEST INFRASTRUCTURE ONLY -- CPU restatement of StemGNN's spectral hot path.

This file restates, step for step, what the reference computes on its hot path
(``/root/reference/models/base_model.py``) as plain functional torch-CPU code
(fp32 or fp64), with torch autograd supplying the reference's implicit backward
(``models/handler.py:164``).  It is the checker for the HIP kernels and the
reported ``cpu_baseline`` ("port") of bench.py.  It is never imported by the
product package ``stemgnn_amd`` and is not a fallback for it.

Pinned against the reference itself: ``tests/golden/make_golden.py`` imports the
real reference (under the 4-item compat shim of ``oracle/ref_shim.py``) in the
build container, runs it on fixed inputs and commits the outputs; the CPU test
suite checks this restatement against those vectors (tests/test_oracle_golden.py)
and, when ``/root/reference`` is present, against the live reference too.
The reference ships no tests / golden vectors of its own.

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# parameter inventory (state_dict contract, models/base_model.py:23-44, 88-101)
# --------------------------------------------------------------------------- #
def param_shapes(units, time_step, multi_layer, horizon, stack_cnt=2):
    """Ordered {state_dict key: shape} in the reference's creation order.

    models/base_model.py:88-101 (Model.__init__) and :23-44 (StockBlockLayer.__init__).
    """
    N, W, m, H = units, time_step, multi_layer, horizon
    Wm, C0, C = W * m, 4 * W, 4 * W * m
    sh = OrderedDict()
    sh["weight_key"] = (N, 1)
    sh["weight_query"] = (N, 1)
    sh["GRU.weight_ih_l0"] = (3 * N, W)
    sh["GRU.weight_hh_l0"] = (3 * N, N)
    sh["GRU.bias_ih_l0"] = (3 * N,)
    sh["GRU.bias_hh_l0"] = (3 * N,)
    for s in range(stack_cnt):
        p = f"stock_block.{s}."
        sh[p + "weight"] = (1, 4, 1, Wm, Wm)
        sh[p + "forecast.weight"] = (Wm, Wm)
        sh[p + "forecast.bias"] = (Wm,)
        sh[p + "forecast_result.weight"] = (W, Wm)
        sh[p + "forecast_result.bias"] = (W,)
        if s == 0:
            sh[p + "backcast.weight"] = (W, Wm)
            sh[p + "backcast.bias"] = (W,)
        sh[p + "backcast_short_cut.weight"] = (W, W)
        sh[p + "backcast_short_cut.bias"] = (W,)
        for g in range(6):
            cin = C0 if g < 2 else C
            for side in ("left", "right"):
                sh[p + f"GLUs.{g}.linear_{side}.weight"] = (C, cin)
                sh[p + f"GLUs.{g}.linear_{side}.bias"] = (C,)
    sh["fc.0.weight"] = (W, W)
    sh["fc.0.bias"] = (W,)
    sh["fc.2.weight"] = (H, W)
    sh["fc.2.bias"] = (H,)
    return sh


def det_state_dict(units, time_step, multi_layer, horizon, seed=0, dtype=torch.float32, stack_cnt=2):
    """Deterministic, version-independent weights of init-like magnitude (test fixture weights)."""
    from .detrand import det_uniform

    sd = OrderedDict()
    for i, (k, shape) in enumerate(param_shapes(units, time_step, multi_layer, horizon, stack_cnt).items()):
        if k in ("weight_key", "weight_query"):
            bound = 1.414 * math.sqrt(6.0 / (shape[0] + 1))
        elif k.endswith(".weight") and len(shape) == 5:
            bound = math.sqrt(3.0) * math.sqrt(2.0 / (2 * shape[-1] * shape[1]))
        elif k.startswith("GRU."):
            bound = 1.0 / math.sqrt(units)
        else:
            fan_in = shape[-1] if len(shape) > 1 else None
            if fan_in is None:  # bias: fan_in of its weight
                wkey = k[: -len("bias")] + "weight"
                fan_in = param_shapes(units, time_step, multi_layer, horizon, stack_cnt)[wkey][-1]
            bound = 1.0 / math.sqrt(fan_in)
        arr = det_uniform(shape, seed * 1000 + i, -bound, bound)
        sd[k] = torch.from_numpy(arr).to(dtype)
    return sd


# --------------------------------------------------------------------------- #
# front: GRU -> self attention -> Laplacian -> Chebyshev
# --------------------------------------------------------------------------- #
def gru_manual(seq, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU(time_step, units) written out (torch.nn.GRU docs; the cell ATen's _VF.gru evaluates):
        r = s(W_ir x + b_ir + W_hr h + b_hr),  z = s(W_iz x + b_iz + W_hz h + b_hz),
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) * n + z * h,   gate order (r, z, n), h_0 = 0.
    seq [S,B,W] -> [S,B,H].  Device-agnostic plain torch ops: lets the LARGE parity cases evaluate the fp64 yardstick
    through torch on the GPU (MIOpen has no fp64 RNN); pinned to _VF.gru on the CPU by tests/test_oracle_golden.py."""
    H = w_hh.shape[1]
    gi = torch.matmul(seq, w_ih.t()) + b_ih
    h = torch.zeros(seq.shape[1], H, dtype=seq.dtype, device=seq.device)
    w_hh_t = w_hh.t()
    outs = []
    for s in range(seq.shape[0]):
        gh = torch.matmul(h, w_hh_t) + b_hh
        r = torch.sigmoid(gi[s, :, :H] + gh[:, :H])
        z = torch.sigmoid(gi[s, :, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[s, :, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        outs.append(h)
    return torch.stack(outs, dim=0)


def gru_front(x, sd):
    """models/base_model.py:137-138.  x [B,W,N] -> GRU over the node axis -> [B,N_seq,N_hid]."""
    N = sd["weight_key"].shape[0]
    flat = [sd["GRU.weight_ih_l0"], sd["GRU.weight_hh_l0"], sd["GRU.bias_ih_l0"], sd["GRU.bias_hh_l0"]]
    seq = x.permute(2, 0, 1).contiguous()  # [N_seq, B, W]
    if x.is_cuda:      # yardstick evaluated on a device (large cases only): written-out cell, see gru_manual
        return gru_manual(seq, *flat).permute(1, 0, 2).contiguous()
    h0 = torch.zeros(1, x.shape[0], N, dtype=x.dtype)
    out, _ = torch._VF.gru(seq, h0, flat, True, 1, 0.0, False, False, False)
    return out.permute(1, 0, 2).contiguous()  # [B, N_seq, N_hid]


def self_graph_attention(gru_out, wk, wq, alpha=0.2, drop_mask=None, drop_p=0.0, kink_pos=None):
    """models/base_model.py:151-162.

    gru_out [B, N_seq, N_hid]; the adjacency's node axis is the GRU *hidden* index (:152).
    ``drop_mask`` ([B,N,N] of 0/1) stands in for nn.Dropout's Bernoulli mask (:161) so
    train-mode results are reproducible; None = eval mode / p=0.
    ``kink_pos`` (bool [B,N,N], optional): which side of the LeakyReLU kink each logit key_i+query_j is taken to be
    on.  The gradient is discontinuous at 0; with B*N*N logits some lie closer to 0 than the fp32 rounding of
    key/query (tools/kink_probe.py), so a test that compares against higher precision passes the decisions of the
    implementation under test instead of letting the sign of a 1e-8 number decide.
    """
    inp = gru_out.permute(0, 2, 1)                       # :152  [B, i=hid, s=seq]
    key = torch.matmul(inp, wk)                          # :154  [B,N,1]
    query = torch.matmul(inp, wq)                        # :155  [B,N,1]
    data = key + query.transpose(1, 2)                   # :156-158  data[b,i,j] = key[b,i] + query[b,j]
    if kink_pos is None:
        data = F.leaky_relu(data, alpha)                 # :159
    else:
        data = torch.where(kink_pos, data, alpha * data)
    att = torch.softmax(data, dim=2)                     # :160
    if drop_mask is not None and drop_p > 0.0:
        att = att * drop_mask / (1.0 - drop_p)           # :161 (inverted dropout)
    return att


def laplacian_from_attention(att):
    """models/base_model.py:140-147.  att [B,N,N] -> (L [N,N], attention_sym [N,N])."""
    A = att.mean(dim=0)                                   # :140
    degree = A.sum(dim=1)                                 # :141 (before symmetrisation)
    A_s = 0.5 * (A + A.T)                                 # :143
    d_hat = 1.0 / (torch.sqrt(degree) + 1e-7)             # :145
    L = d_hat[:, None] * (torch.diag(degree) - A_s) * d_hat[None, :]   # :144-147
    return L, A_s


def cheb_polynomial(L):
    """models/base_model.py:121-134.  [N,N] -> [4,N,N] = [0, L, 2LL, 2L(2LL) - L] (T0 is ZERO, :129)."""
    T0 = torch.zeros_like(L)
    T1 = L
    T2 = 2.0 * (L @ T1) - T0                              # :131
    T3 = 2.0 * (L @ T2) - T1                              # :132
    return torch.stack([T0, T1, T2, T3], dim=0)


def cheb_from_eig(L):
    """North-star eigen route: T_k(L) = U p_k(Lambda) U^T, p = (0, l, 2l^2, 4l^3 - l)  (SURVEY 0-2).
    Same function of L as cheb_polynomial; used to check the HIP eigensolver path."""
    lam, U = torch.linalg.eigh(L.double())
    polys = [torch.zeros_like(lam), lam, 2 * lam ** 2, 4 * lam ** 3 - lam]
    return torch.stack([(U * p[None, :]) @ U.T for p in polys], dim=0).to(L.dtype)


# --------------------------------------------------------------------------- #
# StockBlockLayer
# --------------------------------------------------------------------------- #
def glu(x, sd, prefix):
    """models/base_model.py:12-13."""
    left = F.linear(x, sd[prefix + "linear_left.weight"], sd[prefix + "linear_left.bias"])
    right = F.linear(x, sd[prefix + "linear_right.weight"], sd[prefix + "linear_right.bias"])
    return left * torch.sigmoid(right)


def spe_seq_cell(gfted, sd, prefix):
    """models/base_model.py:46-59.  gfted [B,4,1,N,W] -> [B,4,N,W*multi].

    torch.rfft(x,1,onesided=False) (torch 1.7) == view_as_real(fft.fft(x, dim=-1));
    torch.irfft(z,1,onesided=False) == C2R inverse that reads only bins 0..n/2 of the
    two-sided input == fft.irfft(z[..., :n//2+1], n=n)   (SURVEY 0-6, 8c shim items 1-2).
    """
    B, k, _, N, W = gfted.shape
    x = gfted.reshape(B, -1, N, W)                        # :48
    on_dev = x.is_cuda     # device evaluation of the yardstick (largest parity cases): the two transforms as explicit DFT
    if on_dev:             # matrices instead of library FFT plans (rocFFT compiles kernels at run time: minutes on a fresh box)
        t = torch.arange(W, dtype=x.dtype, device=x.device)
        ang = 2.0 * math.pi * torch.outer(t, t) / W
        ff_real, ff_imag = x @ torch.cos(ang), -(x @ torch.sin(ang))
    else:
        ff = torch.fft.fft(x, dim=-1)                     # :49
        ff_real, ff_imag = ff.real, ff.imag
    real = ff_real.permute(0, 2, 1, 3).reshape(B, N, -1)  # :50  column = k*W + f
    img = ff_imag.permute(0, 2, 1, 3).reshape(B, N, -1)   # :51
    for i in range(3):                                    # :52-54
        real = glu(real, sd, prefix + f"GLUs.{2 * i}.")
        img = glu(img, sd, prefix + f"GLUs.{2 * i + 1}.")
    real = real.reshape(B, N, 4, -1).permute(0, 2, 1, 3)  # :55
    img = img.reshape(B, N, 4, -1).permute(0, 2, 1, 3)    # :56
    n = real.shape[-1]
    if on_dev:             # C2R of bins 0..n/2 written out: y = (1/n) [sum c_f Re cos - sum s_f Im sin]
        h = n // 2
        f = torch.arange(h + 1, dtype=x.dtype, device=x.device)
        tau = torch.arange(n, dtype=x.dtype, device=x.device)
        ang = 2.0 * math.pi * torch.outer(f, tau) / n
        c = torch.full((h + 1,), 2.0, dtype=x.dtype, device=x.device)
        c[0] = 1.0
        sfac = c.clone()
        sfac[0] = 0.0
        if n % 2 == 0:
            c[h] = 1.0
            sfac[h] = 0.0
        return (real[..., : h + 1] @ (c[:, None] * torch.cos(ang)) - img[..., : h + 1] @ (sfac[:, None] * torch.sin(ang))) / n
    z = torch.complex(real, img)[..., : n // 2 + 1]       # :57
    return torch.fft.irfft(z, n=n, dim=-1)                # :58


def stock_block(X, mul_L, sd, s):
    """models/base_model.py:61-75.  X [B,1,N,W], mul_L [4,N,N] -> (forecast [B,N,W], backcast [B,1,N,W] | None)."""
    p = f"stock_block.{s}."
    gfted = torch.matmul(mul_L.unsqueeze(1), X.unsqueeze(1))            # :62-64  [B,4,1,N,W]
    gconv_input = spe_seq_cell(gfted, sd, p).unsqueeze(2)               # :65     [B,4,1,N,Wm]
    igfted = torch.matmul(gconv_input, sd[p + "weight"]).sum(dim=1)     # :66-67  [B,1,N,Wm]
    fsrc = torch.sigmoid(F.linear(igfted, sd[p + "forecast.weight"], sd[p + "forecast.bias"]).squeeze(1))  # :68
    forecast = F.linear(fsrc, sd[p + "forecast_result.weight"], sd[p + "forecast_result.bias"])           # :69
    if s == 0:                                                          # :70-72
        short = F.linear(X, sd[p + "backcast_short_cut.weight"], sd[p + "backcast_short_cut.bias"])
        back = torch.sigmoid(F.linear(igfted, sd[p + "backcast.weight"], sd[p + "backcast.bias"]) - short)
    else:                                                               # :73-74
        back = None
    return forecast, back


# --------------------------------------------------------------------------- #
# whole model
# --------------------------------------------------------------------------- #
def hot_path(gru_out, x, sd, alpha=0.2, drop_mask=None, drop_p=0.0, spectral="cheb", kink_pos=None):
    """Everything after the GRU up to the summed block forecasts (models/base_model.py:139-148, 169-174).

    gru_out [B,N,N] (batch-first, as after :138), x [B,W,N].
    Returns (fsum [B,N,W], attention [N,N], mul_L [4,N,N]).
    """
    att = self_graph_attention(gru_out, sd["weight_key"], sd["weight_query"], alpha, drop_mask, drop_p, kink_pos)
    L, A_s = laplacian_from_attention(att)
    mul_L = cheb_polynomial(L) if spectral == "cheb" else cheb_from_eig(L)
    X = x.unsqueeze(1).permute(0, 1, 3, 2)                # :169  [B,1,N,W]
    f0, X1 = stock_block(X, mul_L, sd, 0)                 # :171-173
    f1, _ = stock_block(X1, mul_L, sd, 1)
    return f0 + f1, A_s, mul_L                            # :174


def model_forward(x, sd, alpha=0.2, drop_mask=None, drop_p=0.0, spectral="cheb", kink_pos=None):
    """models/base_model.py:167-179.  x [B,W,N] -> (forecast [B,H,N] (or [B,1,N] if H==1), attention [N,N])."""
    gru_out = gru_front(x, sd)
    fsum, A_s, _ = hot_path(gru_out, x, sd, alpha, drop_mask, drop_p, spectral, kink_pos)
    y = F.linear(fsum, sd["fc.0.weight"], sd["fc.0.bias"])               # :175
    y = F.leaky_relu(y, 0.01)
    y = F.linear(y, sd["fc.2.weight"], sd["fc.2.bias"])
    if y.shape[-1] == 1:                                                 # :176-177
        return y.unsqueeze(1).squeeze(-1), A_s
    return y.permute(0, 2, 1).contiguous(), A_s                          # :178-179


def loss_and_grads(x, y, sd, **kw):
    """MSE loss (models/handler.py:140,162) + autograd grads of every parameter that receives one.

    Returns (loss, forecast, attention, {key: grad or None}).
    """
    leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    forecast, att = model_forward(x, leaves, **kw)
    loss = F.mse_loss(forecast, y)
    keys = list(leaves.keys())
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    return loss.detach(), forecast.detach(), att.detach(), OrderedDict(zip(keys, grads))


class OracleTrainer:
    """CPU train step of the restated model: zero_grad -> fwd -> MSE -> bwd -> RMSprop(lr, eps=1e-8).

    Follows models/handler.py:126-127,157-166.  Used by bench.py's cpu_baseline leg (kind="port").
    """

    def __init__(self, units, time_step, multi_layer, horizon, lr=1e-4, seed=0, dropout_rate=0.5):
        self.sd = OrderedDict(
            (k, v.clone().requires_grad_(True))
            for k, v in det_state_dict(units, time_step, multi_layer, horizon, seed).items()
        )
        self.p = dropout_rate
        self.opt = torch.optim.RMSprop(list(self.sd.values()), lr=lr, eps=1e-8)

    def step(self, x, y):
        self.opt.zero_grad(set_to_none=True)
        mask = None
        if self.p > 0.0:
            N = x.shape[2]
            mask = (torch.rand(x.shape[0], N, N) >= self.p).to(x.dtype)
        forecast, _ = model_forward(x, self.sd, drop_mask=mask, drop_p=self.p)
        loss = F.mse_loss(forecast, y)
        loss.backward()
        self.opt.step()
        return float(loss.detach())
