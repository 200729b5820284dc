"""ctypes binding of libttrnn.so (the C ABI declared in include/ttrnn.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C tensorized-rnn_amd/csrc``.
There is no CPU fallback: every compute entry point raises if the library is missing.
"""
import contextlib
import ctypes
import os

import torch  # noqa: F401  (must be imported first: libttrnn resolves libamdhip64.so.7 to torch's copy)

TTRNN_MAX_D = 6
TTRNN_F32, TTRNN_BF16 = 0, 1
TTRNN_LSTM, TTRNN_GRU = 0, 1
PHASE_ALL, PHASE_PREPARE, PHASE_RUN = 0, 1, 2
ABI_VERSION = 7
BWD_STATS_COLMAX, BWD_STATS_IN1SUMS, BWD_STATS_ROWS = 1, 2, 4
BWD_STATS_ROWMAX = 4          # stats = [4][G*H] + [B*T]: the rows' maxima behind the four rows

_HERE = os.path.dirname(os.path.abspath(__file__))
# TTRNN_LIB_PATH: developer override (A/B-ing two builds of the library in one session); default = the in-tree build
LIB_PATH = os.environ.get("TTRNN_LIB_PATH") or os.path.join(_HERE, "libttrnn.so")


class TtmDesc(ctypes.Structure):
    """struct ttrnn_ttm"""
    _fields_ = [("d", ctypes.c_int32),
                ("in_modes", ctypes.c_int32 * TTRNN_MAX_D),
                ("out_modes", ctypes.c_int32 * TTRNN_MAX_D),
                ("ranks", ctypes.c_int32 * (TTRNN_MAX_D + 1))]


class RnnDesc(ctypes.Structure):
    """struct ttrnn_rnn_desc"""
    _fields_ = [("cell", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("seq_len", ctypes.c_int32),
                ("input_size", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
                ("has_bias_in", ctypes.c_int32), ("has_bias_hid", ctypes.c_int32),
                ("in_w", TtmDesc), ("hid_w", TtmDesc), ("hid_blocks", ctypes.c_int32)]


class WgradArgs(ctypes.Structure):
    """struct ttrnn_wgrad_args (include/ttrnn.h, ABI 7): operands of the chain weight gradients of one recurrent layer."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "x", "out", "h0", "d_gates_in", "d_gates_hid", "packed_in", "packed_hid", "d_packed_in", "d_packed_hid",
        "d_bias_in", "d_bias_hid", "x_colmax", "h_colmax", "dy_colmax_in", "dy_colmax_hid")]


class LinHints(ctypes.Structure):
    """struct ttrnn_lin_hints"""
    _fields_ = [("x_colmax", ctypes.c_void_p), ("dy_colmax", ctypes.c_void_p), ("xdy_sum", ctypes.c_void_p),
                ("x_period", ctypes.c_int64), ("x_first", ctypes.c_void_p), ("dy_rowmax", ctypes.c_void_p)]


_P = ctypes.c_void_p
_SIGNATURES = {
    "ttrnn_abi_version": (ctypes.c_int, []),
    "ttrnn_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "ttrnn_device_available": (ctypes.c_int, []),
    "ttrnn_device_status": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint), ctypes.c_int, ctypes.c_int]),
    "ttrnn_set_fp32_math": (ctypes.c_int, [ctypes.c_int]),
    "ttrnn_get_fp32_math": (ctypes.c_int, []),
    "ttrnn_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "ttrnn_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "ttrnn_packed_elems": (ctypes.c_int64, [ctypes.POINTER(TtmDesc)]),
    "ttrnn_pack_cores": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64),
                                        ctypes.c_int, _P, _P]),
    "ttrnn_pack_cores2": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), _P,
                                         ctypes.POINTER(TtmDesc), ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), _P,
                                         ctypes.c_int, _P]),
    "ttrnn_unpack_core_grads": (ctypes.c_int, [ctypes.POINTER(TtmDesc), _P, ctypes.POINTER(_P),
                                               ctypes.POINTER(ctypes.c_int64), ctypes.c_int, _P]),
    "ttrnn_ttlinear_workspace": (ctypes.c_size_t, [ctypes.POINTER(TtmDesc), ctypes.c_int64]),
    "ttrnn_ttlinear_forward": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int64, _P, _P, _P, _P,
                                              _P, ctypes.c_size_t, _P]),
    "ttrnn_ttlinear_backward": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int64, _P, _P, _P, _P,
                                               _P, _P, _P, ctypes.c_size_t, _P]),
    "ttrnn_ttlinear_backward_hinted": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int64, _P, _P,
                                                      _P, _P, _P, _P, ctypes.POINTER(LinHints), _P, ctypes.c_size_t, _P]),
    "ttrnn_ttlinear_backward_shift_ok": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                        ctypes.c_int64, ctypes.c_int]),
    "ttrnn_head_workspace": (ctypes.c_size_t, [ctypes.POINTER(TtmDesc), ctypes.c_int64]),
    "ttrnn_head_forward": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int64, _P, _P, _P, _P, _P,
                                          _P, ctypes.c_size_t, _P]),
    "ttrnn_head_backward": (ctypes.c_int, [ctypes.POINTER(TtmDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int64, _P, _P, _P, _P,
                                           _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "ttrnn_rnn_workspace": (ctypes.c_size_t, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_reserve_bytes": (ctypes.c_size_t, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_forward": (ctypes.c_int, [ctypes.POINTER(RnnDesc)] + [_P] * 12 + [ctypes.c_size_t, _P]),
    "ttrnn_rnn_forward_route": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_forward_samples_per_workgroup": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_prepare_supported": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_out_optional": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_forward_phase": (ctypes.c_int, [ctypes.POINTER(RnnDesc), ctypes.c_int] + [_P] * 12 + [ctypes.c_size_t, _P]),
    "ttrnn_rnn_forward_cores_fused": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_forward_cores": (ctypes.c_int, [ctypes.POINTER(RnnDesc), _P, _P, _P,
                                               ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), _P,
                                               ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), _P,
                                               _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "ttrnn_rnn_backward_workspace": (ctypes.c_size_t, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_backward_workspace_ex": (ctypes.c_size_t, [ctypes.POINTER(RnnDesc), ctypes.c_int]),
    "ttrnn_rnn_backward_route": (ctypes.c_int, [ctypes.POINTER(RnnDesc), ctypes.c_int]),
    "ttrnn_rnn_backward": (ctypes.c_int, [ctypes.POINTER(RnnDesc)] + [_P] * 14 + [ctypes.c_size_t, _P]),
    "ttrnn_rnn_backward_stats": (ctypes.c_int, [ctypes.POINTER(RnnDesc)]),
    "ttrnn_rnn_backward_ex": (ctypes.c_int, [ctypes.POINTER(RnnDesc)] + [_P] * 16 + [ctypes.c_size_t, _P]),
    "ttrnn_rnn_wgrad_workspace": (ctypes.c_size_t, [ctypes.POINTER(RnnDesc), ctypes.c_int]),
    "ttrnn_rnn_wgrad": (ctypes.c_int, [ctypes.POINTER(RnnDesc), ctypes.c_int, ctypes.POINTER(WgradArgs), _P, ctypes.c_size_t, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class TtrnnError(RuntimeError):
    pass


def load():
    """Load libttrnn.so, check the ABI version, bind signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TtrnnError(
            "libttrnn.so not found at {} - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C tensorized-rnn_amd/csrc`). There is no CPU fallback for the TT-RNN hot path.".format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.ttrnn_abi_version() != ABI_VERSION:
        raise TtrnnError("libttrnn.so ABI {} != binding ABI {}".format(lib.ttrnn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ttrnn_status_string(status).decode()
        raise TtrnnError("{} failed: {} ({})".format(what, msg, status))


MATH_MODES = {"exact": 0, "split": 1}     # TTRNN_MATH_EXACT / TTRNN_MATH_SPLIT (include/ttrnn.h)


# bumped whenever a library option changes: prepared workspaces (functional.PreparedLayer) are keyed on it
OPTIONS_EPOCH = 0


def set_fp32_math(mode):
    """Select how the kernels multiply fp32 operands (include/ttrnn.h: TTRNN_MATH_*): "exact" (every product an fp32 MFMA
    product) or "split" (the default: each fp32 operand as two fp16 pieces under exact power-of-two scales, three fp16 MFMA
    terms x0w0 + x0w1 + x1w0, fp32 accumulation — three bf16 pieces / six terms in the few kernels whose operand magnitudes
    are unknown).  Process-wide; returns the previous mode."""
    if mode not in MATH_MODES:
        raise ValueError("fp32 math mode must be one of {}".format(sorted(MATH_MODES)))
    global OPTIONS_EPOCH
    prev = get_fp32_math()
    check(load().ttrnn_set_fp32_math(MATH_MODES[mode]), "ttrnn_set_fp32_math")
    if prev != mode:          # (a no-op set — the restore of a `with` block that changed nothing — keeps captured graphs valid)
        OPTIONS_EPOCH += 1
    return prev


def get_fp32_math():
    code = load().ttrnn_get_fp32_math()
    return next(name for name, c in MATH_MODES.items() if c == code)


@contextlib.contextmanager
def fp32_math(mode):
    prev = set_fp32_math(mode)
    try:
        yield
    finally:
        set_fp32_math(prev)


def get_option(name):
    """Current value of a library option (include/ttrnn.h: ttrnn_get_option)."""
    v = ctypes.c_int(0)
    check(load().ttrnn_get_option(name.encode(), ctypes.byref(v)), "ttrnn_get_option({!r})".format(name))
    return v.value


def set_option(name, value):
    """Set a library option; returns the previous value."""
    global OPTIONS_EPOCH
    prev = get_option(name)
    check(load().ttrnn_set_option(name.encode(), int(value)), "ttrnn_set_option({!r}, {})".format(name, value))
    if prev != int(value):
        OPTIONS_EPOCH += 1
    return prev


@contextlib.contextmanager
def option(name, value):
    """`with option("no_gemm", 1): ...` — route switches for A/B runs and route-vs-route parity tests."""
    prev = set_option(name, value)
    try:
        yield
    finally:
        set_option(name, prev)


def device_status(reset=False):
    """Device-side event counters (include/ttrnn.h: ttrnn_device_status; synchronises the current device):
    {"pair_timeouts": ..., "guard_trips": ..., "block_violations": ...}."""
    buf = (ctypes.c_uint * 4)()
    check(load().ttrnn_device_status(buf, 4, 1 if reset else 0), "ttrnn_device_status")
    return {"pair_timeouts": int(buf[0]), "guard_trips": int(buf[1]), "block_violations": int(buf[2])}


EPILOGUES = {None: 0, "log_softmax": 1, "relu_l2norm": 2}     # TTRNN_EPI_*
ROUTES = {0: "valu", 1: "stagewise_mfma", 2: "fused_core", 3: "merged_big", 4: "runtime_mfma"}


def make_ttm(in_modes, out_modes, ranks):
    d = len(in_modes)
    if d > TTRNN_MAX_D or len(out_modes) != d or len(ranks) != d + 1:
        raise ValueError("TT-matrix with d={} cores is not supported (max {})".format(d, TTRNN_MAX_D))
    t = TtmDesc()
    t.d = d
    for k in range(d):
        t.in_modes[k] = int(in_modes[k])
        t.out_modes[k] = int(out_modes[k])
    for k in range(d + 1):
        t.ranks[k] = int(ranks[k])
    return t
