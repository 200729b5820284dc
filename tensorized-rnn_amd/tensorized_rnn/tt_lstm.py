"""Drop-in ``tensorized_rnn.tt_lstm``: the TT-LSTM cell and model.

Constructor signatures and weight layout follow the reference (``tensorized_rnn/tt_lstm.py``): both
``input_weights`` and ``hidden_weights`` are one gates-concatenated ``TTLinear`` of ``4*hidden``
outputs with the mode shapes of ``tt_shape`` (a bias on BOTH, tt_lstm.py:26,39), or — ``is_naive`` —
a ``TTLinearSet`` of four bias-free TTLinears (tt_lstm.py:17-21,30-34).  The factories live in
``_fused.TTWeightsMixin`` (shared with the GRU); ``forward`` is ``lstm.LSTM``'s and runs on libttrnn's
persistent kernels.
"""
from ._fused import TTStackMixin, TTWeightsMixin
from .lstm import LSTM, LSTMCell


class TTLSTMCell(TTWeightsMixin, LSTMCell):
    n_gate = 4
    naive_bias = False

    def __init__(self, input_size, hidden_size, bias, device, n_cores, tt_rank,
                 is_naive=False, new_core=None):
        self._tt_options(n_cores, tt_rank, is_naive, new_core)
        LSTMCell.__init__(self, input_size, hidden_size, bias, device)


class TTLSTM(TTStackMixin, LSTM):
    tt_cell_cls = TTLSTMCell

    def __init__(self, input_size, hidden_size, num_layers, device, n_cores, tt_rank,
                 bias=True, is_naive=False, log_grads=False, new_core=None):
        self._tt_options(n_cores, tt_rank, is_naive, new_core)
        LSTM.__init__(self, input_size, hidden_size, num_layers, device, bias, log_grads=log_grads)
