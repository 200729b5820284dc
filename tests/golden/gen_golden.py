#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Run in the build container only (needs /root/reference, sympy, scipy):
    python tests/golden/gen_golden.py
The reference is imported from /root/reference — never copied.  Only data (inputs, weights,
expected outputs, expected gradients) is written; the GPU box runs the tests from these files.

Fixture families:
  g1_auto_shape.npz   auto_shape(n, d) for every n <= 4096, d in {2,3,4}
  g2_tt_shape.npz     tt_shape(...) for the configurations and variants in scope
  g3_ttlinear_*.npz   TTLinear forward (+ backward) per distinct layer shape
  g4_cell_*.npz       one LSTM / GRU cell step
  g5_seq_*.npz        full sequences (cfg1..cfg5 shapes, reduced batch), incl. init_states
  g6_bwd_*.npz        BPTT gradients
  g7_init_*.npz       seeded construction -> full state_dict (init parity)
  g8_var_*.npz        variants: naive, new_core first/last, bias=False, dense, multi-layer tiny
Every case file holds: meta (JSON), sd/<key> (+ stride/<key>), inputs, expected outputs.
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

REF = '/root/reference'
sys.path.insert(0, REF)
import t3nsor  # noqa: E402
from t3nsor.layers import TTLinear  # noqa: E402
from t3nsor.utils import auto_shape  # noqa: E402
from tensorized_rnn.gru import GRU, TTGRU  # noqa: E402
from tensorized_rnn.lstm import LSTM  # noqa: E402
from tensorized_rnn.rnn_utils import tt_shape  # noqa: E402
from tensorized_rnn.tt_lstm import TTLSTM  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CPU = torch.device('cpu')
torch.set_num_threads(4)


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def sd_arrays(module):
    out = {}
    for k, v in module.state_dict().items():
        out['sd/' + k] = v.detach().contiguous().numpy().copy()
        out['stride/' + k] = np.array(v.stride(), dtype=np.int64)
    return out


# `python tests/golden/gen_golden.py g6_bwd_cfg4_l3 g6_bwd_cfg5` regenerates only the named cases (prefix match)
ONLY = [a for a in sys.argv[1:] if not a.startswith('-')]


def wanted(name):
    return not ONLY or any(name.startswith(o) for o in ONLY)


def save(name, meta, **arrays):
    if not wanted(name):
        return
    arrays = {k: v for k, v in arrays.items() if v is not None}
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024.0))


def build_rnn(meta):
    torch.manual_seed(meta['seed'])
    kind = meta['kind']
    common = dict(input_size=meta['input_size'], hidden_size=meta['hidden_size'],
                  num_layers=meta['num_layers'], device=CPU)
    if kind in ('ttlstm', 'ttgru'):
        cls = TTLSTM if kind == 'ttlstm' else TTGRU
        return quiet(cls, n_cores=meta['n_cores'], tt_rank=meta['tt_rank'], bias=meta.get('bias', True),
                     is_naive=meta.get('is_naive', False), new_core=meta.get('new_core'), **common)
    cls = LSTM if kind == 'lstm' else GRU
    return quiet(cls, bias=meta.get('bias', True), **common)


def scale_weights(model, factor):
    if factor != 1.0:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if 'bias' not in n:
                    p.mul_(factor)


def seq_case(name, meta, B, T, init=False, t_index=None, input_dist='normal', grads=False, weight_scale=1.0):
    """Forward (and optionally backward) of a full module on a [B,T,in] input."""
    if not wanted(name):
        return
    meta = dict(meta, B=B, T=T, init_states=init, input_dist=input_dist, weight_scale=weight_scale)
    model = build_rnn(meta)
    scale_weights(model, weight_scale)
    lstm = meta['kind'] in ('ttlstm', 'lstm')
    g = torch.Generator().manual_seed(meta['seed'] + 7)
    if input_dist == 'normal':
        x = torch.randn(B, T, meta['input_size'], generator=g)
    else:
        x = torch.rand(B, T, meta['input_size'], generator=g)
    H = meta['hidden_size']
    h0 = c0 = None
    states = None
    if init:
        h0 = torch.randn(B, H, generator=g) * 0.5
        c0 = torch.randn(B, H, generator=g) * 0.5
        states = (h0, c0) if lstm else h0
    arrays = sd_arrays(model)
    if grads:
        x.requires_grad_(True)
        if init:
            h0.requires_grad_(True)
            if lstm:
                c0.requires_grad_(True)
    if grads:
        res = model(x, states)
    else:
        with torch.no_grad():
            res = model(x, states)
    if lstm:
        out, (hT, cT) = res
    else:
        out, hT = res
        cT = None
    arrays['x'] = x.detach().numpy()
    if init:
        arrays['h0'] = h0.detach().numpy()
        if lstm:
            arrays['c0'] = c0.detach().numpy()
    if t_index is not None:
        arrays['out_t_index'] = np.array(t_index, dtype=np.int64)
        arrays['out'] = out.detach()[:, t_index, :].numpy()
    else:
        arrays['out'] = out.detach().numpy()
    arrays['hT'] = hT.detach().numpy()
    if cT is not None:
        arrays['cT'] = cT.detach().numpy()
    if grads:
        w_out = torch.randn(out.shape, generator=g)
        w_h = torch.randn(hT.shape, generator=g)
        loss = (out * w_out).sum() + (hT * w_h).sum()
        arrays['w_out'] = w_out.numpy()
        arrays['w_h'] = w_h.numpy()
        if lstm:
            w_c = torch.randn(cT.shape, generator=g)
            loss = loss + (cT * w_c).sum()
            arrays['w_c'] = w_c.numpy()
        loss.backward()
        arrays['loss'] = np.array(loss.item())
        for n, p in model.named_parameters():
            arrays['grad/' + n] = p.grad.detach().contiguous().numpy().copy()
        arrays['grad_x'] = x.grad.numpy()
        if init:
            arrays['grad_h0'] = h0.grad.numpy()
            if lstm:
                arrays['grad_c0'] = c0.grad.numpy()
    save(name, meta, **arrays)


def cell_case(name, meta, B):
    """One cell step with N(0,1) x, h, c."""
    meta = dict(meta, B=B, num_layers=1)
    model = build_rnn(meta)
    lstm = meta['kind'] in ('ttlstm', 'lstm')
    g = torch.Generator().manual_seed(meta['seed'] + 3)
    x = torch.randn(B, meta['input_size'], generator=g)
    h = torch.randn(B, meta['hidden_size'], generator=g)
    c = torch.randn(B, meta['hidden_size'], generator=g)
    arrays = sd_arrays(model)
    with torch.no_grad():
        if lstm:
            hy, cy = model.cell0(x, h, c)
        else:
            hy, cy = model.cell0(x, h), None
    arrays.update(x=x.numpy(), h=h.numpy(), hy=hy.numpy())
    if lstm:
        arrays.update(c=c.numpy(), cy=cy.numpy())
    save(name, meta, **arrays)


def ttlinear_case(name, shape, tt_rank, N=3, seed=1111, bias=True):
    torch.manual_seed(seed)
    d = len(shape[0])
    out_f = int(np.prod(shape[1]))
    lin = quiet(TTLinear, out_features=out_f, shape=shape, bias=bias, auto_shapes=False, d=d, tt_rank=tt_rank)
    g = torch.Generator().manual_seed(seed + 5)
    x = torch.randn(N, int(np.prod(shape[0])), generator=g, requires_grad=True)
    y = lin(x)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    meta = dict(shape=[list(map(int, shape[0])), list(map(int, shape[1]))], tt_rank=tt_rank, N=N, seed=seed,
                bias=bias, ranks=[int(r) for r in lin.weight_t.ranks])
    arrays = sd_arrays(lin)
    arrays.update(x=x.detach().numpy(), y=y.detach().numpy(), w=w.numpy(), grad_x=x.grad.numpy())
    for n, p in lin.named_parameters():
        arrays['grad/' + n] = p.grad.detach().contiguous().numpy().copy()
    save(name, meta, **arrays)


def init_case(name, meta):
    model = build_rnn(meta)
    save(name, meta, **sd_arrays(model))


def main():
    # ---- G1 ------------------------------------------------------------------------------------
    ns, ds, shapes = [], [], []
    for d in (2, 3, 4):
        for n in range(1, 4097):
            s = [int(v) for v in auto_shape(n, d=d)]
            ns.append(n); ds.append(d); shapes.append(s + [0] * (4 - d))
    save('g1_auto_shape', dict(note='auto_shape(n, d, entropy, ascending)'), n=np.array(ns, dtype=np.int32),
         d=np.array(ds, dtype=np.int32), shape=np.array(shapes, dtype=np.int32))

    # ---- G2 ------------------------------------------------------------------------------------
    rows = []
    for (i, h, d, g) in [(1, 128, 2, 4), (128, 128, 2, 4), (1, 256, 3, 4), (256, 256, 3, 4), (1, 256, 3, 3),
                         (256, 256, 3, 3), (40, 256, 3, 4), (1024, 1024, 4, 4), (28, 64, 2, 4), (64, 64, 2, 3),
                         (28, 256, 2, 4), (28, 256, 3, 3), (256, 512, 3, 4), (40, 768, 3, 4), (10, 100, 2, 3)]:
        for nc in (None, 'first', 'last'):
            sh = tt_shape(i, h, d, g, new_core=nc)
            rows.append(dict(in_features=i, hidden=h, n_cores=d, n_gates=g, new_core=nc, shape=sh))
    save('g2_tt_shape', dict(cases=rows))

    # ---- G3: TTLinear per distinct layer shape (cfg1..cfg5, heads) ------------------------------
    lin_shapes = {
        'cfg1_in': ([[1, 1], [16, 32]], 4), 'cfg1_hid': ([[8, 16], [16, 32]], 4),
        'cfg2_in': ([[1, 1, 1], [8, 8, 16]], 8), 'cfg2_hid': ([[4, 8, 8], [8, 8, 16]], 8),
        'cfg3_in': ([[1, 1, 1], [8, 8, 12]], 8), 'cfg3_hid': ([[4, 8, 8], [8, 8, 12]], 8),
        'cfg4_in': ([[2, 4, 5], [8, 8, 16]], 16), 'cfg4_hid': ([[4, 8, 8], [8, 8, 16]], 16),
        'cfg5_hid': ([[4, 4, 8, 8], [8, 8, 8, 8]], 32),
        'head_mnist': ([[4, 8, 8], [1, 2, 5]], 8), 'head_sv': ([[4, 8, 8], [4, 8, 8]], 16),
        'odd': ([[3, 5, 2], [7, 3, 5]], 3), 'd1': ([[12], [20]], 1), 'd5': ([[2, 3, 2, 2, 3], [3, 2, 2, 3, 2]], 5),
    }
    for nm, (shape, r) in lin_shapes.items():
        ttlinear_case('g3_ttlinear_' + nm, shape, r, N=3)
    ttlinear_case('g3_ttlinear_nobias', [[4, 8, 8], [8, 8, 12]], 8, N=5, bias=False)
    ttlinear_case('g3_ttlinear_rows37', [[4, 8, 8], [8, 8, 16]], 8, N=37)

    cfg1 = dict(kind='ttlstm', input_size=1, hidden_size=128, num_layers=1, n_cores=2, tt_rank=4, seed=1111)
    cfg2 = dict(kind='ttlstm', input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8, seed=1111)
    cfg3 = dict(kind='ttgru', input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8, seed=1111)
    cfg4 = dict(kind='ttlstm', input_size=40, hidden_size=256, num_layers=3, n_cores=3, tt_rank=16, seed=11)
    cfg5 = dict(kind='ttlstm', input_size=1024, hidden_size=1024, num_layers=1, n_cores=4, tt_rank=32, seed=1111)

    # ---- G4: single cell step --------------------------------------------------------------------
    cell_case('g4_cell_cfg1', cfg1, 4)
    cell_case('g4_cell_cfg2', cfg2, 4)
    cell_case('g4_cell_cfg3', cfg3, 4)
    cell_case('g4_cell_cfg4', dict(cfg4, input_size=40), 4)
    cell_case('g4_cell_cfg5', cfg5, 2)

    # ---- G5: sequences ----------------------------------------------------------------------------
    seq_case('g5_seq_cfg1', cfg1, B=4, T=32)
    sub = list(range(0, 784, 56)) + [783]
    seq_case('g5_seq_cfg2', cfg2, B=4, T=784, t_index=sub, input_dist='uniform')
    seq_case('g5_seq_cfg3', cfg3, B=4, T=784, t_index=sub, input_dist='uniform')
    seq_case('g5_seq_cfg4', cfg4, B=4, T=16, input_dist='uniform')
    seq_case('g5_seq_cfg5', cfg5, B=2, T=4)
    seq_case('g5_seq_cfg1_init', cfg1, B=3, T=9, init=True)
    seq_case('g5_seq_cfg3_init', cfg3, B=3, T=9, init=True)
    seq_case('g5_seq_cfg4_init', dict(cfg4, num_layers=2), B=3, T=5, init=True)
    # trained-scale weights over a short horizon (saturating gates)
    seq_case('g5_seq_cfg2_scaled', cfg2, B=3, T=12, weight_scale=1.6)
    seq_case('g5_seq_cfg3_scaled', cfg3, B=3, T=12, weight_scale=1.6)

    # ---- G6: backward -----------------------------------------------------------------------------
    seq_case('g6_bwd_cfg1', cfg1, B=4, T=32, grads=True)
    seq_case('g6_bwd_cfg2', cfg2, B=3, T=64, grads=True)
    seq_case('g6_bwd_cfg3', cfg3, B=3, T=24, grads=True, init=True)
    seq_case('g6_bwd_cfg1_init', cfg1, B=3, T=7, grads=True, init=True, weight_scale=1.5)
    seq_case('g6_bwd_cfg4', dict(cfg4, num_layers=2), B=2, T=6, grads=True, input_dist='uniform')
    seq_case('g6_bwd_cfg4_l3', cfg4, B=2, T=6, grads=True, input_dist='uniform')        # the real cfg4: three layers
    seq_case('g6_bwd_cfg5', cfg5, B=2, T=3, grads=True)
    # the reference's own published model family (experiments/speaker_verification/encoder/params_model.py:2-4,14-16:
    # H = 768, one layer, n_cores = 2, rank = 2; result tables d in {2, 4}): shapes whose chain is cheaper than the dense matrix
    spk = dict(kind='ttlstm', input_size=40, hidden_size=768, num_layers=1, n_cores=2, tt_rank=2, seed=11)
    seq_case('g6_bwd_spk', spk, B=4, T=24, grads=True, input_dist='uniform')
    seq_case('g6_bwd_spk_d4r4', dict(spk, n_cores=4, tt_rank=4), B=4, T=16, grads=True, input_dist='uniform')
    seq_case('g6_bwd_spk_gru', dict(spk, kind='ttgru'), B=3, T=12, grads=True, init=True, input_dist='uniform')

    # ---- G7: init parity ---------------------------------------------------------------------------
    init_case('g7_init_cfg1', cfg1)
    init_case('g7_init_cfg4', cfg4)
    init_case('g7_init_cfg3', cfg3)

    # ---- G8: variants ------------------------------------------------------------------------------
    tiny = dict(input_size=28, hidden_size=64, num_layers=2, n_cores=2, tt_rank=3, seed=1111)
    for kind in ('ttlstm', 'ttgru'):
        seq_case('g8_var_%s_tiny' % kind, dict(tiny, kind=kind), B=3, T=6, grads=True)
        seq_case('g8_var_%s_naive' % kind, dict(tiny, kind=kind, is_naive=True), B=3, T=5, grads=True)
        seq_case('g8_var_%s_first' % kind, dict(tiny, kind=kind, new_core='first'), B=3, T=5, grads=True)
        seq_case('g8_var_%s_last' % kind, dict(tiny, kind=kind, new_core='last'), B=3, T=5, grads=True)
        seq_case('g8_var_%s_nobias' % kind, dict(tiny, kind=kind, bias=False), B=3, T=5, grads=True)
    seq_case('g8_var_lstm_dense', dict(tiny, kind='lstm'), B=3, T=6, grads=True, init=True)
    seq_case('g8_var_gru_dense', dict(tiny, kind='gru'), B=3, T=6, grads=True, init=True)
    seq_case('g8_var_ttlstm_b1t1', dict(tiny, kind='ttlstm', num_layers=1), B=1, T=1)


if __name__ == '__main__':
    main()
