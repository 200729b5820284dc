"""Drop-in ``tensorized_rnn.gru``: ``GRUCell`` / ``GRU`` and their TT variants.

API parity with the reference (``tensorized_rnn/gru.py``).  Cell arithmetic (gru.py:33-44, the
cuDNN / PyTorch convention, gate order r, z, n):
    r = sigmoid(in_r + hid_r)          z = sigmoid(in_z + hid_z)
    n = tanh(in_n + r * hid_n)         h' = (1 - z) * n + z * h
where ``in_*`` / ``hid_*`` include their own biases (both dense and TT GRU cells carry a bias on
both weight sets, gru.py:20,23,157-159).  ``forward(input[B,T,in], init_states=None)`` returns
``(outputs[B,T,H], h[B,H])`` (gru.py:118-136).  The time loop runs inside libttrnn's persistent
kernel (see ``_fused.py``).
"""
import torch
from torch import nn

from ._fused import FusedCellMixin, FusedRnnBase, TTStackMixin, TTWeightsMixin


class GRUCell(FusedCellMixin, nn.Module):
    kind = 'gru'

    def __init__(self, input_size, hidden_size, bias, device):
        nn.Module.__init__(self)
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bias = bias
        self.device = device
        self.input_weights = self._create_input_hidden_weights()
        self.hidden_weights = self._create_hidden_hidden_weights()

    def _create_input_hidden_weights(self):
        return nn.Linear(self.input_size, 3 * self.hidden_size, self.bias).to(self.device)

    def _create_hidden_hidden_weights(self):
        return nn.Linear(self.hidden_size, 3 * self.hidden_size, self.bias).to(self.device)

    def forward(self, input, hx):
        """One timestep: (x[B,in], h[B,H]) -> h'."""
        if self._fusable():
            (hy,) = self._fused_step(input, hx)
        else:
            H = self.hidden_size
            gi = self.input_weights(input)
            gh = self.hidden_weights(hx)
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            hy = (1 - z) * n + z * hx
        if hasattr(self, '_h_backward_hook') and hy.requires_grad:
            hy.register_hook(self._h_backward_hook)
        return hy


class GRU(FusedRnnBase):
    kind = 'gru'

    def __init__(self, input_size, hidden_size, num_layers, device, bias=True, log_grads=False):
        super(GRU, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bias = bias
        self.device = device
        self.log_grads = log_grads
        self._build_layers(log_grads)

    def _create_first_layer_cell(self):
        return GRUCell(self.input_size, self.hidden_size, self.bias, self.device)

    def _create_other_layer_cell(self):
        return GRUCell(self.hidden_size, self.hidden_size, self.bias, self.device)

    def init_hidden(self, batch_size):
        return torch.zeros(batch_size, self.hidden_size).to(self.device)

    def forward(self, input, init_states=None, need_outputs=True):
        """
        :param input:       (batch_size, seq_len, input_size)
        :param init_states: optional h (batch_size, hidden_size); seeds every layer.
        :param need_outputs: extension to the reference's signature (as LSTM.forward): False under torch.no_grad() tells the
                 last layer that only the final state is consumed (mnist_classifier.py:52-55 classifies the last step) —
                 `outputs` is then None and the [B, T, H] store is skipped.
        :return: outputs (batch_size, seq_len, hidden_size) of the last layer and its final h.
        """
        if self._needs_stepping():
            h = self.init_hidden(input.shape[0]) if init_states is None else init_states
            outputs, hT, _ = self._forward_stepwise(input, h.to(input.dtype), None)
        else:
            outputs, hT = self._forward_fused(input, init_states, None, need_outputs)
        return outputs, hT


class TTGRUCell(TTWeightsMixin, GRUCell):
    n_gate = 3

    def __init__(self, input_size, hidden_size, bias, device, n_cores, tt_rank,
                 is_naive=False, new_core=None):
        self._tt_options(n_cores, tt_rank, is_naive, new_core)
        GRUCell.__init__(self, input_size, hidden_size, bias, device)


class TTGRU(TTStackMixin, GRU):
    tt_cell_cls = TTGRUCell

    def __init__(self, input_size, hidden_size, num_layers, device, n_cores, tt_rank,
                 bias=True, is_naive=False, log_grads=False, new_core=None):
        self._tt_options(n_cores, tt_rank, is_naive, new_core)
        GRU.__init__(self, input_size, hidden_size, num_layers, device, bias, log_grads=log_grads)
