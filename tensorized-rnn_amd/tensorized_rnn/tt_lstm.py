"""Drop-in ``tensorized_rnn.tt_lstm``: TT weight factories for the LSTM.

Same constructor signatures and weight layout as the reference (``tensorized_rnn/tt_lstm.py``):
both ``input_weights`` and ``hidden_weights`` are one gates-concatenated ``TTLinear`` of
``4*hidden`` outputs with the mode shapes of ``tt_shape`` (a bias on BOTH, tt_lstm.py:26,39), or —
``is_naive=True`` — a ``TTLinearSet`` of four bias-free TTLinears (tt_lstm.py:17-21,30-34).
The forward pass is inherited from ``lstm.LSTM`` and runs on libttrnn's fused kernels.
"""
from t3nsor.layers import TTLinear

from .lstm import LSTM, LSTMCell
from .rnn_utils import tt_shape
from .tt_linearset import TTLinearSet


class TTLSTMCell(LSTMCell):
    def __init__(self, input_size, hidden_size, bias, device, n_cores, tt_rank,
                 is_naive=False, new_core=None):
        self.n_cores = n_cores
        self.tt_rank = tt_rank
        self.is_naive = is_naive
        self.new_core = new_core
        self.n_gate = 4
        super().__init__(input_size, hidden_size, bias, device)

    def _tt_weights(self, in_features):
        if self.is_naive:
            layer = TTLinearSet(in_features=in_features, out_features=self.hidden_size, n_gates=self.n_gate,
                                bias=False, auto_shapes=True, d=self.n_cores, tt_rank=self.tt_rank)
        else:
            shape = tt_shape(in_features, self.hidden_size, self.n_cores, self.n_gate, new_core=self.new_core)
            layer = TTLinear(out_features=self.n_gate * self.hidden_size, shape=shape, bias=self.bias,
                             auto_shapes=False, d=self.n_cores, tt_rank=self.tt_rank)
        return layer.to(self.device)

    def _create_input_hidden_weights(self):
        return self._tt_weights(self.input_size)

    def _create_hidden_hidden_weights(self):
        return self._tt_weights(self.hidden_size)


class TTLSTM(LSTM):
    def __init__(self, input_size, hidden_size, num_layers, device, n_cores, tt_rank,
                 bias=True, is_naive=False, log_grads=False, new_core=None):
        assert new_core in [None, 'first', 'last']
        self.n_cores = n_cores
        self.tt_rank = tt_rank
        self.is_naive = is_naive
        self.new_core = new_core
        super().__init__(input_size, hidden_size, num_layers, device, bias, log_grads=log_grads)

    def _make_cell(self, in_features):
        return TTLSTMCell(in_features, self.hidden_size, self.bias, self.device, n_cores=self.n_cores,
                          tt_rank=self.tt_rank, is_naive=self.is_naive, new_core=self.new_core)

    def _create_first_layer_cell(self):
        return self._make_cell(self.input_size)

    def _create_other_layer_cell(self):
        return self._make_cell(self.hidden_size)
