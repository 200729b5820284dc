"""ctypes loader for the plain-C oracle (oracle/ttrnn_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libttrnn_oracle.so")
MAX_D = 6


class OracleTtm(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int), ("J", ctypes.c_int * MAX_D), ("I", ctypes.c_int * MAX_D),
                ("R", ctypes.c_int * (MAX_D + 1)), ("core", ctypes.c_void_p * MAX_D), ("bias", ctypes.c_void_p)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


class Ttm(object):
    """cores: list of float32 arrays with logical shape (R_k, I_k, J_k, R_{k+1}); bias: array or None."""

    def __init__(self, cores, bias=None):
        self.cores = [np.ascontiguousarray(c, dtype=np.float32) for c in cores]
        self.bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        t = OracleTtm()
        t.d = len(self.cores)
        for k, c in enumerate(self.cores):
            t.R[k], t.I[k], t.J[k] = c.shape[0], c.shape[1], c.shape[2]
            t.core[k] = c.ctypes.data
        t.R[t.d] = self.cores[-1].shape[3]
        t.bias = self.bias.ctypes.data if self.bias is not None else None
        self.struct = t
        self.in_size = int(np.prod([c.shape[2] for c in self.cores]))
        self.out_size = int(np.prod([c.shape[1] for c in self.cores]))


def ttlinear(w, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty((x.shape[0], w.out_size), dtype=np.float32)
    rc = load().oracle_ttlinear(ctypes.byref(w.struct), _fp(x), _fp(y), ctypes.c_int(x.shape[0]))
    assert rc == 0
    return y


def lstm_layer(w_in, w_hid, x, h0=None, c0=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, _ = x.shape
    H = w_hid.in_size
    out = np.empty((B, T, H), dtype=np.float32)
    hT = np.empty((B, H), dtype=np.float32)
    cT = np.empty((B, H), dtype=np.float32)
    h0 = None if h0 is None else np.ascontiguousarray(h0, dtype=np.float32)
    c0 = None if c0 is None else np.ascontiguousarray(c0, dtype=np.float32)
    rc = load().oracle_lstm_layer(ctypes.byref(w_in.struct), ctypes.byref(w_hid.struct), B, T, _fp(x), _fp(h0),
                                  _fp(c0), _fp(out), _fp(hT), _fp(cT))
    assert rc == 0, rc
    return out, hT, cT


def gru_layer(w_in, w_hid, x, h0=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, T, _ = x.shape
    H = w_hid.in_size
    out = np.empty((B, T, H), dtype=np.float32)
    hT = np.empty((B, H), dtype=np.float32)
    h0 = None if h0 is None else np.ascontiguousarray(h0, dtype=np.float32)
    rc = load().oracle_gru_layer(ctypes.byref(w_in.struct), ctypes.byref(w_hid.struct), B, T, _fp(x), _fp(h0),
                                 _fp(out), _fp(hT))
    assert rc == 0, rc
    return out, hT
