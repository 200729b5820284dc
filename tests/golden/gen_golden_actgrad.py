#!/usr/bin/env python3
"""G10 fixtures: what the reference's ActivGradLogger records (tensorized_rnn/rnn_utils.py:42-226) for one minibatch of a
TT-LSTM / TT-GRU built with log_grads=True — per layer and timestep the batch mean of ||h_t||^2, log ||h_t||^2 (and c_t), and
of the squared norms of the gradients arriving at h_t / c_t — produced by RUNNING THE REFERENCE on CPU in the build container.

    python tests/golden/gen_golden_actgrad.py
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from tensorized_rnn.gru import TTGRU  # noqa: E402
from tensorized_rnn.rnn_utils import ActivGradLogger  # noqa: E402
from tensorized_rnn.tt_lstm import TTLSTM  # noqa: E402


def main():
    for tag, cls, meta in (
            ('ttlstm', TTLSTM, dict(kind='ttlstm', input_size=12, hidden_size=64, num_layers=2, n_cores=2, tt_rank=3, seed=1111)),
            ('ttgru', TTGRU, dict(kind='ttgru', input_size=12, hidden_size=64, num_layers=2, n_cores=2, tt_rank=3, seed=1111)),
            ('ttlstm_cfg2', TTLSTM, dict(kind='ttlstm', input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8, seed=1111))):
        ActivGradLogger.all_loggers.clear()
        torch.manual_seed(meta['seed'])
        with redirect_stdout(io.StringIO()):
            model = cls(meta['input_size'], meta['hidden_size'], meta['num_layers'], torch.device('cpu'), n_cores=meta['n_cores'],
                        tt_rank=meta['tt_rank'], log_grads=True)
        g = torch.Generator().manual_seed(3)
        B, T = 5, 9
        x = torch.randn(B, T, meta['input_size'], generator=g)
        w = torch.randn(B, T, meta['hidden_size'], generator=g)
        res = model(x)
        out = res[0]
        (out * w).sum().backward()
        arrays = {}
        for k, v in model.state_dict().items():
            arrays['sd/' + k] = v.detach().contiguous().numpy().copy()
            arrays['stride/' + k] = np.array(v.stride(), dtype=np.int64)
        for name, lg in ActivGradLogger.all_loggers.items():
            for q in ('act', 'log_act', 'grad', 'log_grad'):
                arrays['log/%s/%s' % (name, q)] = torch.stack(list(getattr(lg, q))).numpy()
        arrays.update(x=x.numpy(), w=w.numpy(), out=out.detach().numpy())
        np.savez_compressed(os.path.join(OUT, 'g10_actgrad_%s.npz' % tag), meta=np.array(json.dumps(dict(meta, B=B, T=T))), **arrays)
        print('wrote g10_actgrad_' + tag, sorted(ActivGradLogger.all_loggers))


if __name__ == '__main__':
    main()
