"""Error of a TT-LSTM recurrence (cfg2 shape) under different emulations of the fused-stage product, against float64.

A float64 run is the truth; every other run rounds h, the stage-1 result, the pre-activations and the state to fp32 and
multiplies the fused core with the stage-1 result in one of these ways:
    f32       genuine fp32 products, fp32 accumulation
    bf16x3    three bf16 pieces per operand, six terms (the split mode of the other kernels)
    f16x2_3   two fp16 pieces per operand under a power-of-two scale, terms x0w0 + x0w1 + x1w0 (fused-core LSTM forward)
    f16x2_4   the same plus x1w1

    python tools/split_precision_sim.py [weight_scale] [timesteps]

CPU only; needs nothing but torch and this repo's modules (parameters of a freshly initialised TTLSTM).
"""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
torch.manual_seed(0)
from tensorized_rnn.tt_lstm import TTLSTM  # noqa: E402

wscale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
T_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 784
with contextlib.redirect_stdout(io.StringIO()):
    mod = TTLSTM(1, 256, 1, torch.device("cpu"), n_cores=3, tt_rank=8)
sd = {k: v.detach().clone().double() for k, v in mod.state_dict().items()}
# cores (R_k, I_k, J_k, R_{k+1}) of the hidden-to-hidden TT-matrix; the in = 1 projection as a dense column
G0, G1, G2 = [sd["cell0.hidden_weights.parameters.%d" % k] * wscale ** (1 / 3) for k in range(3)]
bias_h = sd["cell0.hidden_weights.bias"]
I0, J0, I1, J1, I2, J2 = G0.shape[1], G0.shape[2], G1.shape[1], G1.shape[2], G2.shape[1], G2.shape[2]
R2 = G2.shape[0]
W10 = torch.einsum('aijb,bklc->ikjlc', G0, G1).reshape(I0 * I1, J0 * J1 * R2)      # [(i0 i1)][(j0 j1 r2)]
G2m = G2[:, :, :, 0]                                                                   # [r2][i2][j2]
A0, A1, A2 = [sd["cell0.input_weights.parameters.%d" % k] for k in range(3)]
Win = torch.einsum('aijb,bklc,cmnd->ikm', A0, A1, A2).reshape(-1)                      # in = 1: [1024]
bin_ = sd["cell0.input_weights.bias"]
B = 8
x = torch.rand(B, T_steps, dtype=torch.float64)

def rnd(t, dt):
    return t.to(dt).to(torch.float64)

def split(t, dt, n):
    out, r = [], t.clone()
    for _ in range(n):
        p = rnd(r, dt); out.append(p); r = r - p
    return out

def matmul32(a, b):            # fp32 accumulate of exactly representable products
    return (a.float() @ b.float()).double()

def product(mode, W, Tm):
    """W [M][K] (float64 values that are fp32-representable), Tm [B][K][N] -> [B][M][N]"""
    if mode == "f64": return W @ Tm
    if mode == "f32": return matmul32(W, Tm)
    if mode == "bf16x3":
        w, t = split(W, torch.bfloat16, 3), split(Tm, torch.bfloat16, 3)
        terms = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
    elif mode in ("f16x2_3", "f16x2_4"):
        w, t = split(W * 2.0 ** 6, torch.float16, 2), split(Tm * 2.0 ** 6, torch.float16, 2)
        terms = [(1, 0), (0, 1), (0, 0)] if mode == "f16x2_3" else [(1, 1), (1, 0), (0, 1), (0, 0)]
    acc = None
    for (i, j) in terms:
        p = matmul32(w[i], t[j])
        acc = p if acc is None else (acc.float() + p.float()).double()
    if mode.startswith("f16"): acc = acc * 2.0 ** -12
    return acc

def run(mode):
    h = torch.zeros(B, 256, dtype=torch.float64); c = h.clone()
    outs = []
    W = W10 if mode == "f64" else rnd(W10, torch.float32)
    for t in range(T_steps):
        hh = h if mode == "f64" else rnd(h, torch.float32)
        Tm = torch.einsum('rij,bklj->bklri', G2m, hh.reshape(B, J0, J1, J2)).reshape(B, J0 * J1 * R2, I2)   # [B][(j0 j1 r2)][i2]
        if mode != "f64": Tm = rnd(Tm, torch.float32)
        pre = product(mode, W, Tm).reshape(B, 1024) + bias_h + x[:, t:t + 1] * Win + bin_
        if mode != "f64": pre = rnd(pre, torch.float32)
        i, f, g, o = pre[:, :256], pre[:, 256:512], pre[:, 512:768], pre[:, 768:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        if mode != "f64": c, h = rnd(c, torch.float32), rnd(h, torch.float32)
        outs.append(h)
    return torch.stack(outs, 1)

ref = run("f64")
print("max|W10| %.3g  max|h| %.3g  std(out) %.3g" % (float(W10.abs().max()), float(ref.abs().max()), float(ref.std())))
for mode in ("f32", "bf16x3", "f16x2_4", "f16x2_3"):
    o = run(mode)
    e = (o - ref).abs()
    print("%-8s max err %.3g  rms err %.3g" % (mode, float(e.max()), float(e.pow(2).mean().sqrt())))
