#!/usr/bin/env python3
"""G9 head fixtures: outputs and gradients of the reference's OWN caller classes — MNIST_Classifier
(experiments/digit_classification/mnist_classifier.py:13-57) and SpeakerEncoder.forward
(experiments/speaker_verification/encoder/speaker_encoder.py:17-91) — run on CPU in the build container.

    python tests/golden/gen_golden_heads.py

The reference is imported from /root/reference (never copied).  Its two model files import sibling modules that only
exist to fix sys.path (`context`) or that pull in audio / plotting dependencies absent here; those names are satisfied
with empty stand-in modules so that the class definitions themselves — the code under test — load unchanged."""
import importlib.util
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import tensorized_rnn  # noqa: E402  (the reference's package)


def load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    with redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def sd_arrays(module):
    out = {}
    for k, v in module.state_dict().items():
        out['sd/' + k] = v.detach().contiguous().numpy().copy()
        out['stride/' + k] = np.array(v.stride(), dtype=np.int64)
    return out


def main():
    ctx = types.ModuleType('context')
    ctx.tensorized_rnn = tensorized_rnn
    sys.modules['context'] = ctx
    mc = load('ref_mnist_classifier', os.path.join(REF, 'experiments/digit_classification/mnist_classifier.py'))

    # ---- pMNIST classifier, cfg1 flags (--tt --ncores 2 --ttrank 4, hidden 128), LSTM and GRU ---------------------------
    for tag, gru in (('lstm', False), ('gru', True)):
        torch.manual_seed(1111)
        with redirect_stdout(io.StringIO()):
            model = mc.MNIST_Classifier(1, 10, 128, 1, torch.device('cpu'), tt=True, gru=gru, n_cores=2, tt_rank=4)
        g = torch.Generator().manual_seed(5)
        x = torch.rand(6, 20, 1, generator=g)
        target = torch.randint(0, 10, (6,), generator=g)
        out = model(x)
        loss = torch.nn.functional.nll_loss(out, target)
        loss.backward()
        arrays = sd_arrays(model)
        arrays.update(x=x.numpy(), target=target.numpy(), out=out.detach().numpy(), loss=np.array(loss.item()))
        for n, p in model.named_parameters():
            arrays['grad/' + n] = p.grad.detach().contiguous().numpy().copy()
        meta = dict(model='MNIST_Classifier', gru=gru, input_size=1, output_size=10, hidden_size=128, num_layers=1, n_cores=2,
                    tt_rank=4, seed=1111)
        np.savez_compressed(os.path.join(OUT, 'g9_head_mnist_%s.npz' % tag), meta=np.array(json.dumps(meta)), **arrays)
        print('wrote g9_head_mnist_' + tag)

    # ---- speaker encoder: TT-LSTM -> TTLinear -> ReLU -> L2 norm -----------------------------------------------------------
    enc_pkg = types.ModuleType('encoder')
    enc_pkg.__path__ = [os.path.join(REF, 'experiments/speaker_verification/encoder')]
    sys.modules['encoder'] = enc_pkg
    ectx = types.ModuleType('encoder.context')
    ectx.tensorized_rnn = tensorized_rnn
    sys.modules['encoder.context'] = ectx
    sys.path.insert(0, os.path.join(REF, 'experiments/speaker_verification'))
    se = load('encoder.speaker_encoder', os.path.join(REF, 'experiments/speaker_verification/encoder/speaker_encoder.py'),
              package='encoder')
    for tag, gru, layers, H in (('lstm', False, 2, 256), ('gru', True, 1, 128)):
        torch.manual_seed(11)
        with redirect_stdout(io.StringIO()):
            model = se.SpeakerEncoder(40, H, layers, 256, torch.device('cpu'), torch.device('cpu'), compression='tt', n_cores=3,
                                      rank=4, use_gru=gru)
        g = torch.Generator().manual_seed(6)
        x = torch.rand(6, 12, 40, generator=g)
        w = torch.randn(6, 256, generator=g)
        emb = model(x)
        loss = (emb * w).sum()
        loss.backward()
        arrays = sd_arrays(model)
        arrays.update(x=x.numpy(), w=w.numpy(), out=emb.detach().numpy(), loss=np.array(loss.item()))
        for n, p in model.named_parameters():
            if p.grad is not None:
                arrays['grad/' + n] = p.grad.detach().contiguous().numpy().copy()
        meta = dict(model='SpeakerEncoder', use_gru=gru, mel_n_channels=40, hidden_size=H, num_layers=layers, embedding_size=256,
                    n_cores=3, rank=4, seed=11)
        np.savez_compressed(os.path.join(OUT, 'g9_head_sv_%s.npz' % tag), meta=np.array(json.dumps(meta)), **arrays)
        print('wrote g9_head_sv_' + tag)


if __name__ == '__main__':
    main()
