"""ActivGradLogger on the fused path: the per-timestep statistics the reference records through forward /
tensor hooks (tensorized_rnn/rnn_utils.py:127-171,217-226), pinned by fixtures produced by the reference itself
(tests/golden/gen_golden_actgrad.py)."""
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tensorized-rnn_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "g10_actgrad_*.npz")))


def _load(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = json.loads(str(d["meta"]))
    sd = {}
    for k in d.files:
        if k.startswith("sd/"):
            t = torch.from_numpy(d[k])
            st = tuple(int(v) for v in d["stride/" + k[3:]])
            if t.dim() > 0 and st != t.stride():
                buf = torch.empty_strided(t.shape, st, dtype=t.dtype)
                buf.copy_(t)
                t = buf
            sd[k[3:]] = t
    return d, meta, sd


@pytest.mark.parametrize("name", CASES)
def test_fixture_is_consistent_with_the_oracle(name):
    """The last layer's hidden activations ARE `outputs`; the oracle's forward reproduces them and therefore the recorded
    activation statistics of that layer (the gradient statistics are checked on the device against the same fixture)."""
    from oracle import ttrnn_oracle as O
    d, meta, sd = _load(name)
    L = meta["num_layers"]
    layers, _ = O.layers_from_state_dict(sd, L)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        out = (O.lstm_forward(layers, x) if meta["kind"] == "ttlstm" else O.gru_forward(layers, x))[0]
    assert float((out - torch.from_numpy(d["out"])).abs().max()) <= 1e-6
    n = out.square().sum(2)
    ref = d["log/hidden_%d/act" % (L - 1)]
    assert np.abs(n.mean(0).numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    refl = d["log/hidden_%d/log_act" % (L - 1)]
    assert np.abs(n.log().mean(0).numpy() - refl).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_log_grads_on_the_fused_path_matches_reference_records(name):
    from tensorized_rnn.rnn_utils import ActivGradLogger
    d, meta, sd = _load(name)
    dev = torch.device("cuda:0")
    ActivGradLogger.all_loggers.clear()
    import contextlib
    import io
    from tensorized_rnn.gru import TTGRU
    from tensorized_rnn.tt_lstm import TTLSTM
    cls = TTLSTM if meta["kind"] == "ttlstm" else TTGRU
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(meta["input_size"], meta["hidden_size"], meta["num_layers"], dev, n_cores=meta["n_cores"], tt_rank=meta["tt_rank"],
                log_grads=True)
    m.load_state_dict(sd, strict=True)
    assert not m._needs_stepping()                              # log_grads=True stays on the fused sequence kernels
    x = torch.from_numpy(d["x"]).to(dev)
    w = torch.from_numpy(d["w"]).to(dev)
    out = m(x)[0]
    (out * w).sum().backward()
    assert float((out.detach().cpu() - torch.from_numpy(d["out"])).abs().max()) <= 1e-5
    keys = [k for k in d.files if k.startswith("log/")]
    assert keys
    for k in keys:
        _, lname, q = k.split("/")
        lg = ActivGradLogger.all_loggers[lname]
        got = torch.stack(list(getattr(lg, q))).float().cpu().numpy()
        ref = d[k]
        assert got.shape == ref.shape == (meta["T"],), (k, got.shape)
        if q.startswith("log"):
            assert np.abs(got - ref).max() <= 2e-4, (k, np.abs(got - ref).max())          # log of a squared norm
        else:
            assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-12), (k, got, ref)
    # the bookkeeping of the reference's training loop still works on these records
    ActivGradLogger.end_minibatch()
    ActivGradLogger.end_epoch()
    logs = ActivGradLogger.get_logs()
    assert logs[("hidden_0", "grad")].shape == (1, meta["T"])
    ActivGradLogger.all_loggers.clear()
