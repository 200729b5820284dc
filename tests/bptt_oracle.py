"""Test helper: gradients of the ORACLE (oracle/ttrnn_oracle.py, float64) for a masked sequence loss

    loss = sum_b (out[b] * w[b]).sum() + (hT * v_h).sum() + (cT * v_c).sum()

at sizes where one autograd graph over all T steps does not fit in host memory (cfg5: 1 024 steps of an H = 1 024 chain
keep ~3 MB of float64 intermediates per sample and step).  For a one-layer model the sequence is cut into segments: a
no-grad pass stores the state at every segment boundary, then the segments are replayed last to first, each with its own
small autograd graph, the state gradient of segment s + 1 entering segment s as a linear term (hT * d_h).sum().  That is
BPTT by hand on top of the oracle's own forward functions — same arithmetic, bounded memory.  Multi-layer models (the
reference shares ONE init state between layers and returns the last layer's state only, tensorized_rnn/lstm.py:120-135,
so a segment cannot be restarted from the oracle's outputs) go through plain autograd.

tests/test_oracle_golden.py checks on CPU that the segmented replay equals plain autograd.
"""
import torch

from oracle import ttrnn_oracle as O


def masked_loss_grads(*args, **kwargs):
    """_masked_loss_grads on at most 8 torch threads: the path is ~40 tiny ATen ops per step, and a 128-core host
    oversubscribed with one thread per core runs it 13x SLOWER than 8 threads do (BENCH_r03 cpu_baseline: 72 vs 969 steps/s)."""
    n = torch.get_num_threads()
    torch.set_num_threads(max(1, min(8, n)))
    try:
        return _masked_loss_grads(*args, **kwargs)
    finally:
        torch.set_num_threads(n)


def _masked_loss_grads(kind, sd, num_layers, x, w, v_h, v_c=None, h0=None, c0=None, seg=64, dtype=torch.float64):
    """kind: 'ttlstm' | 'ttgru' (or 'lstm' / 'gru').  sd: reference-keyed state_dict (CPU tensors).  x [n, T, in], w [n, T, H],
    v_h / v_c [n, H] (v_c LSTM only), h0 / c0 [n, H] or None (zeros, no gradient returned).
    Returns dict(params={key: grad}, dx, dh0, dc0, out, hT, cT) — float64 CPU tensors."""
    lstm = kind in ("ttlstm", "lstm")
    layers, leaves = O.layers_from_state_dict(sd, num_layers, requires_grad=True, dtype=dtype)
    x = x.to(dtype)
    w = w.to(dtype)
    n, T, _ = x.shape
    H = w.shape[2]
    has_state = h0 is not None
    h_init = h0.to(dtype) if has_state else torch.zeros(n, H, dtype=dtype)
    c_init = (c0.to(dtype) if c0 is not None else torch.zeros(n, H, dtype=dtype)) if lstm else None
    v_h = v_h.to(dtype)
    v_c = v_c.to(dtype) if (lstm and v_c is not None) else (torch.zeros(n, H, dtype=dtype) if lstm else None)

    def fwd(xs, h, c):
        if lstm:
            out, (hT, cT) = O.lstm_forward(layers, xs, (h, c))
            return out, hT, cT
        out, hT = O.gru_forward(layers, xs, h)
        return out, hT, None

    if num_layers > 1 or seg >= T:
        xs = x.clone().requires_grad_(True)
        hs = h_init.clone().requires_grad_(True)
        cs = c_init.clone().requires_grad_(True) if lstm else None
        out, hT, cT = fwd(xs, hs, cs)
        obj = (out * w).sum() + (hT * v_h).sum()
        if lstm:
            obj = obj + (cT * v_c).sum()
        obj.backward()
        return dict(params={k: t.grad for k, t in leaves.items()}, dx=xs.grad, dh0=hs.grad if has_state else None,
                    dc0=cs.grad if (lstm and has_state) else None, out=out.detach(), hT=hT.detach(),
                    cT=cT.detach() if lstm else None)

    cuts = list(range(0, T, seg)) + [T]
    states = []
    outs = []
    h, c = h_init, c_init
    with torch.no_grad():
        for s, e in zip(cuts[:-1], cuts[1:]):
            states.append((h, c))
            out, h, c = fwd(x[:, s:e], h, c)
            outs.append(out)
    hT_final, cT_final = h, c
    dx = torch.zeros_like(x)
    dh, dc = v_h, v_c
    for i in reversed(range(len(cuts) - 1)):
        s, e = cuts[i], cuts[i + 1]
        xs = x[:, s:e].clone().requires_grad_(True)
        hs = states[i][0].clone().requires_grad_(True)
        cs = states[i][1].clone().requires_grad_(True) if lstm else None
        out, hT, cT = fwd(xs, hs, cs)
        obj = (out * w[:, s:e]).sum() + (hT * dh).sum()
        if lstm:
            obj = obj + (cT * dc).sum()
        obj.backward()                                  # parameter gradients accumulate in the leaves
        dx[:, s:e] = xs.grad
        dh = hs.grad
        dc = cs.grad if lstm else None
    return dict(params={k: t.grad for k, t in leaves.items()}, dx=dx, dh0=dh if has_state else None,
                dc0=dc if (lstm and has_state) else None, out=torch.cat(outs, 1), hT=hT_final, cT=cT_final)
