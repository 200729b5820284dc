"""CPU check of the generic kernel bodies (tensorized-rnn_amd/csrc/ttrnn_core.h) through the serial
host executor in tests/hostemu/ — packing, chain stages, cell update, BPTT and weight gradients —
against the golden fixtures.  Test infrastructure only: nothing here is reachable from the product.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from golden_io import Case, case_names
from ttrnn_hip._lib import RnnDesc, TtmDesc, make_ttm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "hostemu", "libttrnn_hostemu.so")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "hostemu", "hostemu.cpp")
    hdr = os.path.join(ROOT, "tensorized-rnn_amd", "csrc", "ttrnn_core.h")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "tensorized-rnn_amd", "csrc"), "-o", SO, src])
    lib = ctypes.CDLL(SO)
    lib.hostemu_packed_elems.restype = ctypes.c_int64
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


class Tt(object):
    """A TT-matrix rebuilt from a golden state_dict with the reference's physical stride layout."""

    def __init__(self, emu, cores_logical, strides, bias):
        self.emu = emu
        self.d = len(cores_logical)
        self.views = []
        for c, st in zip(cores_logical, strides):
            # physical buffer in (R, J, I, R') order when the fixture says so; else contiguous
            order = np.argsort([-s for s in st], kind="stable")
            phys = np.ascontiguousarray(np.transpose(c, order))
            view = np.transpose(phys, np.argsort(order))
            assert view.shape == c.shape and np.array_equal(view, c)
            self.views.append(view)
        c = cores_logical
        self.desc = make_ttm([x.shape[2] for x in c], [x.shape[1] for x in c],
                             [x.shape[0] for x in c] + [c[-1].shape[3]])
        self.in_f = int(np.prod([x.shape[2] for x in c]))
        self.out_f = int(np.prod([x.shape[1] for x in c]))
        self.bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        n = emu.hostemu_packed_elems(ctypes.byref(self.desc))
        assert n > 0
        self.packed = np.zeros(n, dtype=np.float32)
        ptrs, st = self._args(self.views)
        assert emu.hostemu_pack(ctypes.byref(self.desc), ptrs, st, _p(self.packed)) == 0

    def _args(self, arrays):
        ptrs = (ctypes.c_void_p * self.d)(*[a.ctypes.data for a in arrays])
        st = (ctypes.c_int64 * (4 * self.d))()
        for k, a in enumerate(arrays):
            for q in range(4):
                st[4 * k + q] = a.strides[q] // 4
        return ptrs, st

    def unpack(self, packed_grad):
        grads = [np.zeros_like(v) for v in self.views]   # order='K': same (transposed) layout as the parameters
        ptrs, st = self._args(grads)
        assert self.emu.hostemu_unpack(ctypes.byref(self.desc), _p(packed_grad), ptrs, st) == 0
        return grads

    def backward(self, x, dy, nb=2, nthr=48, need_dx=True):
        n = x.shape[0]
        dx = np.zeros((n, self.in_f), dtype=np.float32) if need_dx else None
        dpk = np.zeros_like(self.packed)
        db = np.zeros(self.out_f, dtype=np.float32)
        rc = self.emu.hostemu_ttlinear_backward(ctypes.byref(self.desc), ctypes.c_int64(n), _p(self.packed),
                                                _p(np.ascontiguousarray(x)), _p(np.ascontiguousarray(dy)), _p(dx),
                                                _p(dpk), _p(db), nb, nthr)
        assert rc == 0
        return dx, self.unpack(dpk), db


def _tt_from_sd(emu, case, prefix):
    sd, strides = case.state_dict(), case.strides()
    if prefix + "weight" in sd:     # dense nn.Linear as a single (1, out, in, 1) core
        w = sd[prefix + "weight"].numpy()
        core = w.reshape(1, w.shape[0], w.shape[1], 1)
        b = sd[prefix + "bias"].numpy() if prefix + "bias" in sd else None
        return Tt(emu, [core], [(core.size, w.shape[1], 1, 1)], b), [prefix + "weight"]
    cores, sts, keys = [], [], []
    k = 0
    while prefix + "parameters.%d" % k in sd:
        cores.append(sd[prefix + "parameters.%d" % k].numpy())
        sts.append(strides[prefix + "parameters.%d" % k])
        keys.append(prefix + "parameters.%d" % k)
        k += 1
    b = sd[prefix + "bias"].numpy() if prefix + "bias" in sd else None
    return Tt(emu, cores, sts, b), keys


def _maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@pytest.mark.parametrize("name", case_names("g3_ttlinear_"))
def test_hostemu_ttlinear(emu, name):
    case = Case(name)
    tt, keys = _tt_from_sd(emu, case, "")
    x = case.arr["x"]
    y = np.zeros((x.shape[0], tt.out_f), dtype=np.float32)
    assert emu.hostemu_ttlinear_forward(ctypes.byref(tt.desc), ctypes.c_int64(x.shape[0]), _p(tt.packed), _p(tt.bias),
                                        _p(x), _p(y), 2, 37) == 0
    assert _maxabs(y, case.arr["y"]) <= 2e-6
    dx, dcores, db = tt.backward(x, case.arr["w"])
    scale = lambda e: 1e-5 * max(float(np.abs(e).max()), 1e-6) + 1e-7
    assert _maxabs(dx, case.arr["grad_x"]) <= scale(case.arr["grad_x"])
    for key, g in zip(keys, dcores):
        assert _maxabs(g, case.arr["grad/" + key]) <= scale(case.arr["grad/" + key]), key
    if "grad/bias" in case.arr:
        assert _maxabs(db, case.arr["grad/bias"]) <= scale(case.arr["grad/bias"])


SEQ_CASES = [n for n in case_names("g5_seq_") + case_names("g6_bwd_") + case_names("g8_var_") if "naive" not in n
             and n not in ("g5_seq_cfg5",)]


@pytest.mark.parametrize("name", SEQ_CASES)
def test_hostemu_rnn(emu, name):
    case = Case(name)
    meta = case.meta
    lstm = meta["kind"] in ("ttlstm", "lstm")
    G = 4 if lstm else 3
    H, L = meta["hidden_size"], meta["num_layers"]
    x = np.ascontiguousarray(case.arr["x"])
    B, T, _ = x.shape
    h0, c0 = case.arr.get("h0"), case.arr.get("c0")
    want_grads = "loss" in case.arr
    layers = []
    seq = x
    for l in range(L):
        w_in, k_in = _tt_from_sd(emu, case, "cell%d.input_weights." % l)
        w_hid, k_hid = _tt_from_sd(emu, case, "cell%d.hidden_weights." % l)
        d = RnnDesc()
        d.cell, d.dtype, d.batch, d.seq_len = (0 if lstm else 1), 0, B, T
        d.input_size, d.hidden_size = w_in.in_f, H
        d.has_bias_in, d.has_bias_hid = int(w_in.bias is not None), int(w_hid.bias is not None)
        d.in_w, d.hid_w = w_in.desc, w_hid.desc
        out = np.zeros((B, T, H), dtype=np.float32)
        hT = np.zeros((B, H), dtype=np.float32)
        cT = np.zeros((B, H), dtype=np.float32)
        reserve = np.zeros((B, T, (5 if lstm else 4) * H), dtype=np.float32)
        rc = emu.hostemu_rnn_forward(ctypes.byref(d), _p(seq), _p(h0), _p(c0), _p(w_in.packed), _p(w_in.bias),
                                     _p(w_hid.packed), _p(w_hid.bias), _p(out), _p(hT), _p(cT), _p(reserve), 2, 40)
        assert rc == 0
        layers.append(dict(desc=d, w_in=w_in, w_hid=w_hid, k_in=k_in, k_hid=k_hid, x=seq, out=out, reserve=reserve,
                           prefix="cell%d." % l))
        seq = out
    exp_out = case.arr["out"]
    got = seq[:, case.arr["out_t_index"], :] if "out_t_index" in case.arr else seq
    assert _maxabs(got, exp_out) <= 2e-6
    assert _maxabs(hT, case.arr["hT"]) <= 2e-6
    if lstm:
        assert _maxabs(cT, case.arr["cT"]) <= 2e-6
    if not want_grads:
        return
    # ---- backward: reverse-time kernel body per layer, then the two TTLinear weight passes --------
    d_out = np.ascontiguousarray(case.arr["w_out"])
    d_hT = np.ascontiguousarray(case.arr["w_h"])
    d_cT = np.ascontiguousarray(case.arr["w_c"]) if lstm else None
    d_h0_tot = np.zeros((B, H), dtype=np.float32)
    d_c0_tot = np.zeros((B, H), dtype=np.float32)
    scale = lambda e: 2e-5 * max(float(np.abs(e).max()), 1e-6) + 1e-7
    for l in reversed(range(L)):
        ly = layers[l]
        dg_in = np.zeros((B, T, G * H), dtype=np.float32)
        dg_hid = dg_in if lstm else np.zeros((B, T, G * H), dtype=np.float32)
        d_h0 = np.zeros((B, H), dtype=np.float32)
        d_c0 = np.zeros((B, H), dtype=np.float32)
        rc = emu.hostemu_rnn_backward(ctypes.byref(ly["desc"]), _p(ly["out"]), _p(h0), _p(c0), _p(ly["w_hid"].packed),
                                      _p(ly["reserve"]), _p(d_out), _p(d_hT), _p(d_cT), _p(dg_in), _p(dg_hid),
                                      _p(d_h0), _p(d_c0), 2, 33)
        assert rc == 0
        d_h0_tot += d_h0
        d_c0_tot += d_c0
        dx, dcin, dbin = ly["w_in"].backward(ly["x"].reshape(B * T, -1), dg_in.reshape(B * T, -1))
        first = h0 if h0 is not None else np.zeros((B, H), dtype=np.float32)
        hprev = np.concatenate([first[:, None, :], ly["out"][:, :-1]], axis=1).reshape(B * T, H)
        _, dchid, dbhid = ly["w_hid"].backward(hprev, dg_hid.reshape(B * T, -1), need_dx=False)
        for key, g in list(zip(ly["k_in"], dcin)) + list(zip(ly["k_hid"], dchid)):
            exp = case.arr["grad/" + key]
            assert _maxabs(g.reshape(exp.shape), exp) <= scale(exp), key
        for nm, db, w in (("input_weights", dbin, ly["w_in"]), ("hidden_weights", dbhid, ly["w_hid"])):
            if w.bias is not None:
                exp = case.arr["grad/%s%s.bias" % (ly["prefix"], nm)]
                assert _maxabs(db, exp) <= scale(exp), nm
        # gradient w.r.t. this layer's input sequence feeds the layer below; only the top layer
        # receives d_hT / d_cT
        d_out = dx.reshape(B, T, -1)
        d_hT = None
        d_cT = None
    assert _maxabs(d_out, case.arr["grad_x"]) <= scale(case.arr["grad_x"])
    if "grad_h0" in case.arr:
        assert _maxabs(d_h0_tot, case.arr["grad_h0"]) <= scale(case.arr["grad_h0"])
    if "grad_c0" in case.arr:
        assert _maxabs(d_c0_tot, case.arr["grad_c0"]) <= scale(case.arr["grad_c0"])
