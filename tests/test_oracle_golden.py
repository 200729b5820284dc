"""Pins the oracle (oracle/ttrnn_oracle.py and oracle/ttrnn_oracle.c) against the golden fixtures
generated from the reference itself (tests/golden/gen_golden.py).  CPU only.

Tolerances: the torch oracle repeats the reference's ATen op sequence, so forward results are
expected bit-close (<= 1e-6 abs); the C oracle accumulates in double (<= 2e-6 abs).  Gradients
<= 1e-5 relative to the tensor's max magnitude.
"""
import numpy as np
import pytest
import torch

from golden_io import Case, case_names
from oracle import c_oracle
from oracle import ttrnn_oracle as O

torch.set_num_threads(4)


def _maxabs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def _run_oracle(case, requires_grad=False):
    meta = case.meta
    layers, leaves = O.layers_from_state_dict(case.state_dict(), meta["num_layers"], requires_grad)
    x = case.tensor("x").clone()
    h0, c0 = case.tensor("h0"), case.tensor("c0")
    if requires_grad:
        x.requires_grad_(True)
        if h0 is not None:
            h0 = h0.clone().requires_grad_(True)
        if c0 is not None:
            c0 = c0.clone().requires_grad_(True)
    lstm = meta["kind"] in ("ttlstm", "lstm")
    if lstm:
        init = None if h0 is None else (h0, c0)
        out, (hT, cT) = O.lstm_forward(layers, x, init)
    else:
        out, hT = O.gru_forward(layers, x, h0)
        cT = None
    return dict(out=out, hT=hT, cT=cT, x=x, h0=h0, c0=c0, leaves=leaves)


def _check_forward(case, res, tol):
    out = res["out"].detach()
    if "out_t_index" in case.arr:
        out = out[:, torch.from_numpy(case.arr["out_t_index"]), :]
    assert _maxabs(out, case.arr["out"]) <= tol
    assert _maxabs(res["hT"].detach(), case.arr["hT"]) <= tol
    if res["cT"] is not None:
        assert _maxabs(res["cT"].detach(), case.arr["cT"]) <= tol


@pytest.mark.parametrize("name", case_names("g5_seq_") + case_names("g8_var_"))
def test_torch_oracle_sequences(name):
    case = Case(name)
    with torch.no_grad():
        res = _run_oracle(case)
    _check_forward(case, res, 1e-6)


@pytest.mark.parametrize("name", case_names("g6_bwd_") + [n for n in case_names("g8_var_") if "b1t1" not in n])
def test_torch_oracle_gradients(name):
    case = Case(name)
    res = _run_oracle(case, requires_grad=True)
    _check_forward(case, res, 1e-6)
    loss = (res["out"] * case.tensor("w_out")).sum() + (res["hT"] * case.tensor("w_h")).sum()
    if res["cT"] is not None:
        loss = loss + (res["cT"] * case.tensor("w_c")).sum()
    loss.backward()
    assert abs(loss.item() - float(case.arr["loss"])) <= 1e-4 * max(1.0, abs(float(case.arr["loss"])))

    def close(got, exp, what):
        scale = max(float(np.abs(exp).max()), 1e-6)
        assert _maxabs(got, exp) <= 1e-5 * scale + 1e-7, what

    for key, g in case.grads().items():
        if key.split(".")[1] in ("input_weights", "hidden_weights") and ".gate" in key and ".gates." not in key:
            continue  # gate{i} aliases gates.{i}; the oracle keeps the ModuleList copy
        close(res["leaves"][key].grad, g.numpy(), key)
    close(res["x"].grad, case.arr["grad_x"], "grad_x")
    if "grad_h0" in case.arr:
        close(res["h0"].grad, case.arr["grad_h0"], "grad_h0")
    if "grad_c0" in case.arr:
        close(res["c0"].grad, case.arr["grad_c0"], "grad_c0")


@pytest.mark.parametrize("name", case_names("g3_ttlinear_"))
def test_oracles_ttlinear(name):
    case = Case(name)
    sd = case.state_dict()
    cores = [sd["parameters.%d" % k] for k in range(len(case.meta["shape"][0]))]
    bias = sd.get("bias")
    x = case.tensor("x")
    y = O.ttlinear(cores, bias, x)
    assert _maxabs(y, case.arr["y"]) <= 1e-6
    yc = c_oracle.ttlinear(c_oracle.Ttm([c.numpy() for c in cores], None if bias is None else bias.numpy()), x.numpy())
    assert _maxabs(yc, case.arr["y"]) <= 2e-6


@pytest.mark.parametrize("name", case_names("g4_cell_"))
def test_oracles_cell(name):
    case = Case(name)
    layers, _ = O.layers_from_state_dict(case.state_dict(), 1)
    w_in, w_hid = layers[0]
    x, h = case.tensor("x"), case.tensor("h")
    cin = c_oracle.Ttm([c.numpy() for c in w_in[1]], w_in[2].numpy())
    chid = c_oracle.Ttm([c.numpy() for c in w_hid[1]], w_hid[2].numpy())
    with torch.no_grad():
        if case.meta["kind"] == "ttlstm":
            c = case.tensor("c")
            hy, cy = O.lstm_cell(w_in, w_hid, x, h, c)
            assert _maxabs(hy, case.arr["hy"]) <= 1e-6 and _maxabs(cy, case.arr["cy"]) <= 1e-6
            out, hT, cT = c_oracle.lstm_layer(cin, chid, x.numpy()[:, None, :], h.numpy(), c.numpy())
            assert _maxabs(hT, case.arr["hy"]) <= 5e-6 and _maxabs(cT, case.arr["cy"]) <= 5e-6
        else:
            hy = O.gru_cell(w_in, w_hid, x, h)
            assert _maxabs(hy, case.arr["hy"]) <= 1e-6
            out, hT = c_oracle.gru_layer(cin, chid, x.numpy()[:, None, :], h.numpy())
            assert _maxabs(hT, case.arr["hy"]) <= 5e-6


def _c_layers(case):
    layers, _ = O.layers_from_state_dict(case.state_dict(), case.meta["num_layers"])
    res = []
    for w_in, w_hid in layers:
        if w_in[0] != "tt" or w_hid[0] != "tt":
            return None
        res.append((c_oracle.Ttm([c.numpy() for c in w_in[1]], None if w_in[2] is None else w_in[2].numpy()),
                    c_oracle.Ttm([c.numpy() for c in w_hid[1]], None if w_hid[2] is None else w_hid[2].numpy())))
    return res


@pytest.mark.parametrize("name", case_names("g5_seq_") + ["g8_var_ttlstm_first", "g8_var_ttgru_last",
                                                           "g8_var_ttlstm_nobias", "g8_var_ttgru_tiny"])
def test_c_oracle_sequences(name):
    """The plain-C restatement, layer by layer, against the reference's outputs."""
    case = Case(name)
    layers = _c_layers(case)
    seq = case.arr["x"]
    h0, c0 = case.arr.get("h0"), case.arr.get("c0")
    lstm = case.meta["kind"] == "ttlstm"
    for w_in, w_hid in layers:
        if lstm:
            seq, hT, cT = c_oracle.lstm_layer(w_in, w_hid, seq, h0, c0)
        else:
            seq, hT = c_oracle.gru_layer(w_in, w_hid, seq, h0)
    out = seq
    if "out_t_index" in case.arr:
        out = out[:, case.arr["out_t_index"], :]
    assert _maxabs(out, case.arr["out"]) <= 1e-5
    assert _maxabs(hT, case.arr["hT"]) <= 1e-5
    if lstm:
        assert _maxabs(cT, case.arr["cT"]) <= 1e-5


@pytest.mark.parametrize("kind", ["ttlstm", "ttgru"])
def test_segmented_bptt_helper_equals_plain_autograd(kind):
    """tests/bptt_oracle.py (the float64 oracle gradients of the full-size backward parity tests): replaying the sequence in
    segments with the state gradient handed from segment to segment equals one autograd graph over all steps."""
    import contextlib
    import io
    from bptt_oracle import masked_loss_grads
    from golden_io import build_module
    torch.manual_seed(5)
    meta = dict(kind=kind, input_size=3, hidden_size=16, num_layers=1, n_cores=2, tt_rank=3)
    with contextlib.redirect_stdout(io.StringIO()):
        m = build_module(meta, torch.device("cpu"))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    n, T, H = 3, 23, 16
    x, w = torch.randn(n, T, 3), torch.randn(n, T, H)
    vh, vc, h0, c0 = torch.randn(n, H), torch.randn(n, H), torch.randn(n, H) * 0.3, torch.randn(n, H) * 0.3
    for with_state in (True, False):
        kw = dict(h0=h0, c0=c0) if with_state else {}
        a = masked_loss_grads(kind, sd, 1, x, w, vh, vc, seg=T + 1, **kw)
        b = masked_loss_grads(kind, sd, 1, x, w, vh, vc, seg=5, **kw)
        for k in a["params"]:
            assert (a["params"][k] - b["params"][k]).abs().max() <= 1e-12 * max(1.0, float(a["params"][k].abs().max())), k
        for k in ("dx", "dh0", "dc0", "out", "hT", "cT"):
            if a[k] is None:
                assert b[k] is None
            else:
                assert (a[k] - b[k]).abs().max() <= 1e-12 * max(1.0, float(a[k].abs().max())), k
