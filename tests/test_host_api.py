"""CPU tests of the host side: shape selection, init parity, state_dict contract, and that
libttrnn.so loads and exports every symbol include/ttrnn.h declares (no compute calls)."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from golden_io import Case, build_module

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_auto_shape_matches_reference_table():
    from t3nsor.utils import auto_shape
    case = Case("g1_auto_shape")
    n, d, shape = case.arr["n"], case.arr["d"], case.arr["shape"]
    for i in range(len(n)):
        assert auto_shape(int(n[i]), d=int(d[i])) == [int(v) for v in shape[i][:d[i]]], (n[i], d[i])


def test_auto_shape_known_answers():
    from t3nsor.utils import auto_shape
    # SURVEY.md 8(a6) probed values
    assert auto_shape(1, 3) == [1, 1, 1]
    assert auto_shape(40, 3) == [2, 4, 5]
    assert auto_shape(768, 3) == [8, 8, 12]
    assert auto_shape(1024, 3) == [8, 8, 16]
    assert auto_shape(1024, 4) == [4, 4, 8, 8]
    assert auto_shape(4096, 4) == [8, 8, 8, 8]
    with pytest.raises(ValueError):
        auto_shape(12, 3, criterion="bogus")
    with pytest.raises(ValueError):
        auto_shape(12, 3, mode="bogus")


def test_tt_shape_matches_reference():
    from tensorized_rnn.rnn_utils import tt_shape
    for row in Case("g2_tt_shape").meta["cases"]:
        got = tt_shape(row["in_features"], row["hidden"], row["n_cores"], row["n_gates"], new_core=row["new_core"])
        assert got == row["shape"], row
    with pytest.raises(AssertionError):
        tt_shape(4, 4, 2, 4, new_core="middle")


@pytest.mark.parametrize("name", ["g7_init_cfg1", "g7_init_cfg4", "g7_init_cfg3"])
def test_init_parity_bit_exact(name):
    """Same seed -> bit-identical cores / biases, same keys, same strides as the reference."""
    case = Case(name)
    torch.manual_seed(case.meta["seed"])
    m = build_module(case.meta, torch.device("cpu"))
    sd = m.state_dict()
    exp, strides = case.state_dict(), case.strides()
    assert list(sd.keys()) == list(exp.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(exp[k].shape), k
        assert v.stride() == strides[k], k
        assert torch.equal(v, exp[k]), k


def test_param_counts():
    # SURVEY.md 8(a11) probed values
    cpu = torch.device("cpu")
    mk = lambda **kw: build_module(dict(num_layers=1, **kw), cpu)
    assert mk(kind="ttlstm", input_size=1, hidden_size=128, n_cores=2, tt_rank=4).param_count() == 3776
    assert mk(kind="ttlstm", input_size=1, hidden_size=256, n_cores=3, tt_rank=8).param_count() == 8128
    assert mk(kind="ttgru", input_size=1, hidden_size=256, n_cores=3, tt_rank=8).param_count() == 7328
    m = build_module(dict(kind="ttlstm", input_size=40, hidden_size=256, num_layers=3, n_cores=3, tt_rank=16), cpu)
    assert m.param_count() == 110592
    assert [n for n, _ in m.named_children()] == ["cell0", "cell1", "cell2"]
    assert len(m._all_layers) == 3


def test_state_dict_roundtrip_and_variants():
    cpu = torch.device("cpu")
    for name in ("g8_var_ttlstm_naive", "g8_var_ttgru_first", "g8_var_ttlstm_nobias", "g8_var_lstm_dense"):
        case = Case(name)
        m = build_module(case.meta, cpu)
        res = m.load_state_dict(case.state_dict(), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        strides = case.strides()
        for k, v in m.state_dict().items():
            assert v.stride() == strides[k], (name, k)
    # naive variant registers gate{i} and gates.{i} (tt_linearset.py:23,25)
    keys = Case("g8_var_ttlstm_naive").state_dict().keys()
    assert any(".gate0." in k for k in keys) and any(".gates.0." in k for k in keys)


def test_ttlinear_constructor_contract(capsys):
    from t3nsor.layers import TTLinear
    lin = TTLinear(in_features=256, out_features=10, bias=True, auto_shapes=True, d=3, tt_rank=8)
    assert "Created TTLinear layer with input shape: [4, 8, 8]. output shape: [1, 2, 5]" in capsys.readouterr().out
    assert lin.shape == [[4, 8, 8], [1, 2, 5]]
    assert lin.weight_t.ranks == [1, 8, 8, 1] and lin.weight_t.ndims == 3
    assert lin.weight_t.raw_shape == [[1, 2, 5], [4, 8, 8]]
    assert sum(p.numel() for p in lin.parameters()) == 1386     # 1376 core weights + 10 bias
    assert torch.allclose(lin.bias, torch.full((10,), 1e-3))
    assert all(getattr(p, "is_tt", False) for p in lin.weight_t.tt_cores)
    with pytest.raises(ValueError):
        TTLinear(auto_shapes=True)
    with pytest.raises(ValueError):
        TTLinear(auto_shapes=False)
    # dense equivalent of the stored layout (weight_t is the transposed TT-matrix)
    dense = lin.weight_t.full().detach()
    assert dense.shape == (10, 256)
    from oracle import ttrnn_oracle as O
    x = torch.randn(5, 256)
    y = O.ttlinear([c.detach() for c in lin.weight_t.tt_cores], lin.bias.detach(), x)
    assert torch.allclose(y, x @ dense.t() + lin.bias.detach(), atol=1e-5)


def test_no_cpu_fallback():
    from ttrnn_hip import TtrnnError
    with contextlib.redirect_stdout(io.StringIO()):
        m = build_module(dict(kind="ttgru", input_size=4, hidden_size=8, num_layers=1, n_cores=2, tt_rank=2),
                         torch.device("cpu"))
    with pytest.raises(TtrnnError):
        m(torch.zeros(2, 3, 4))
    with pytest.raises(TtrnnError):
        m.cell0.input_weights(torch.zeros(2, 4))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ttrnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ttrnn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ttrnn_hip import _lib
    lib = _lib.load()                       # raises if libttrnn.so is not built
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert lib.ttrnn_abi_version() == _lib.ABI_VERSION == 7
    assert lib.ttrnn_status_string(0) == b"ok"
    assert b"workspace" in lib.ttrnn_status_string(-4)


def test_descriptor_validation_without_gpu():
    from ttrnn_hip import _lib
    from ttrnn_hip.functional import RnnLayerSpec, TTSpec
    lib = _lib.load()
    good = _lib.make_ttm([4, 8, 8], [8, 8, 16], [1, 8, 8, 1])
    assert lib.ttrnn_packed_elems(ctypes.byref(good)) == 2 * 5376
    bad = _lib.make_ttm([4, 8, 8], [8, 8, 16], [2, 8, 8, 1])       # R_0 must be 1
    assert lib.ttrnn_packed_elems(ctypes.byref(bad)) == -1
    zero = _lib.make_ttm([4, 0, 8], [8, 8, 16], [1, 8, 8, 1])
    assert lib.ttrnn_packed_elems(ctypes.byref(zero)) == -1
    with pytest.raises(ValueError):
        _lib.make_ttm([2] * 7, [2] * 7, [1] + [2] * 6 + [1])
    spec = RnnLayerSpec("lstm", 1, 256, TTSpec([1, 1, 1], [8, 8, 16], [1, 8, 8, 1]),
                        TTSpec([4, 8, 8], [8, 8, 16], [1, 8, 8, 1]), True, True)
    d = spec.desc(64, 784, 0)
    assert lib.ttrnn_rnn_reserve_bytes(ctypes.byref(d)) == 64 * 784 * 5 * 256 * 4      # gates [.][4] + cell states [.]
    # cfg2 runs on the shape-specialised kernel; input_size == 1 -> only two hoisted rows [2][H][4] + unit inputs
    # two unit rows of gate inputs + the unit input rows + the fused-core fragments of the f10 kernel (4 x 8 x 3 KB)
    assert lib.ttrnn_rnn_workspace(ctypes.byref(d)) == 2 * 256 * 4 * 4 + 256 + 4 * 8 * 3 * 1024
    wide = RnnLayerSpec("lstm", 256, 256, TTSpec([4, 8, 8], [8, 8, 16], [1, 8, 8, 1]),
                        TTSpec([4, 8, 8], [8, 8, 16], [1, 8, 8, 1]), True, True).desc(64, 784, 0)
    # gate inputs fp32 [B][T][H][4] + the fused-core fragments of the hidden AND of the (hidden-shaped) input matrix
    # + the dense-GEMM K-in: identity rows [in][in], dense W_in [in][4H] (fp32), its 16-bit planes (room for three) and
    # the two-piece fp16 GEMM's scales (one fp32 per column of W_in + one per row of x)
    assert lib.ttrnn_rnn_workspace(ctypes.byref(wide)) == (64 * 784 * 256 * 4 * 4 + 2 * 4 * 8 * 3 * 1024 +
                                                           256 * 256 * 4 + 256 * 1024 * 4 + 3 * 256 * 1024 * 2 +
                                                           1024 * 4 + 64 * 784 * 4)
    tiny = RnnLayerSpec("gru", 28, 64, TTSpec([4, 7], [12, 16], [1, 3, 1]), TTSpec([8, 8], [12, 16], [1, 3, 1]),
                        True, True).desc(3, 6, 0)
    # no shape-specialised kernel: the runtime-shape MFMA route (ttrnn_g2.hip), whose workspace holds the hoisted input
    # projection, the merged cores and their fragment streams; the any-shape VALU kernels keep everything in LDS
    assert lib.ttrnn_rnn_forward_route(ctypes.byref(tiny)) == 4 and lib.ttrnn_rnn_forward_route(ctypes.byref(d)) == 2
    assert lib.ttrnn_rnn_workspace(ctypes.byref(tiny)) >= 3 * 6 * 64 * 4 * 4
    import ttrnn_hip
    with ttrnn_hip.option("no_g2", 1):
        assert lib.ttrnn_rnn_forward_route(ctypes.byref(tiny)) == 0
        assert lib.ttrnn_rnn_workspace(ctypes.byref(tiny)) == 0
    # reverse-time workspace: a caller that passes no d_state pays for its own route only; the unconditional
    # query covers both kinds of call
    for desc in (d, wide, tiny):
        own, with_state = (lib.ttrnn_rnn_backward_workspace_ex(ctypes.byref(desc), w) for w in (0, 1))
        assert lib.ttrnn_rnn_backward_workspace(ctypes.byref(desc)) == max(own, with_state)
    own, with_state = (lib.ttrnn_rnn_backward_workspace_ex(ctypes.byref(d), w) for w in (0, 1))
    assert lib.ttrnn_rnn_backward_route(ctypes.byref(d), 0) == 2 and lib.ttrnn_rnn_backward_route(ctypes.byref(d), 1) != 2
    assert own > 0 and with_state > 0
    big = RnnLayerSpec("lstm", 1024, 1024, TTSpec([4, 8, 8, 4], [8, 8, 8, 8], [1, 32, 32, 32, 1]),
                       TTSpec([4, 8, 8, 4], [8, 8, 8, 8], [1, 32, 32, 32, 1]), True, True).desc(128, 1024, 0)
    own, with_state = (lib.ttrnn_rnn_backward_workspace_ex(ctypes.byref(big), w) for w in (0, 1))
    print("cfg5-class reverse-time workspace: own route", own, "with d_state", with_state)
    assert 0 < own <= lib.ttrnn_rnn_backward_workspace(ctypes.byref(big))
    with pytest.raises(ValueError):
        RnnLayerSpec("gru", 1, 256, TTSpec([1, 1, 1], [8, 8, 16], [1, 8, 8, 1]),
                     TTSpec([4, 8, 8], [8, 8, 16], [1, 8, 8, 1]), True, True)
    # NULL pointers are rejected before anything is launched
    assert lib.ttrnn_rnn_forward(ctypes.byref(d), *([None] * 12), 0, None) == -2


def test_fp32_math_mode_switch_without_gpu():
    """ttrnn_set_fp32_math / ttrnn_get_fp32_math (include/ttrnn.h): pure host state, no device needed."""
    import ttrnn_hip
    from ttrnn_hip import _lib
    lib = _lib.load()
    start = ttrnn_hip.get_fp32_math()
    assert start in ("split", "exact")
    try:
        assert ttrnn_hip.set_fp32_math("exact") == start
        assert ttrnn_hip.get_fp32_math() == "exact" and lib.ttrnn_get_fp32_math() == 0
        with ttrnn_hip.fp32_math("split"):
            assert lib.ttrnn_get_fp32_math() == 1
        assert ttrnn_hip.get_fp32_math() == "exact"            # restored by the context manager
        assert lib.ttrnn_set_fp32_math(7) == -3                # TTRNN_ERR_UNSUPPORTED, mode unchanged
        assert ttrnn_hip.get_fp32_math() == "exact"
        with pytest.raises(ValueError):
            ttrnn_hip.set_fp32_math("tf32")
    finally:
        ttrnn_hip.set_fp32_math(start)
