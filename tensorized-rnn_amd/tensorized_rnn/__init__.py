"""Drop-in ``tensorized_rnn`` package: TT-LSTM / TT-GRU modules whose forward / backward run on
libttrnn.so (hand-written HIP kernels for AMD MI355X).  Put the directory that contains this
package (``tensorized-rnn_amd/``) on ``sys.path`` exactly as the reference's experiments do with
the reference checkout (``experiments/*/context.py``)."""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
