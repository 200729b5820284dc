"""Kept for callers that imported the example module: the GE2E similarity / loss now lives in the product package
(`ttrnn_hip.ge2e`)."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorized-rnn_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from ttrnn_hip.ge2e import (eer, ge2e_loss, ge2e_loss_data_parallel, similarity_matrix)  # noqa: E402,F401
