"""Drop-in ``t3nsor`` package (hot-path subset) backed by libttrnn.so on AMD MI355X.

Exports the names the reference's ``t3nsor/__init__.py:1-16`` exposes that the TT-LSTM / TT-GRU
path uses; symbols the reference exports only for unused code paths (TensorTrainBatch, TTEmbedding,
gather_rows, to_tt_*, tensor_ones/zeros, matrix_zeros, ind2sub) are intentionally absent
(out of scope).
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

from . import utils  # noqa: E402,F401
from .tensor_train import TensorTrain  # noqa: E402,F401
from .initializers import random_matrix, matrix_with_random_cores, glorot_initializer  # noqa: E402,F401
from .ops import tt_dense_matmul, transpose  # noqa: E402,F401
from .layers import TTLinear  # noqa: E402,F401
