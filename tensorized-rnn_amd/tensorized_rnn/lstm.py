"""Drop-in ``tensorized_rnn.lstm``: ``LSTMCell`` and the multi-layer ``LSTM`` whose forward is the
path ``TTLSTM`` inherits.

API parity with the reference (``tensorized_rnn/lstm.py``): constructor signatures, attribute
names (``cell{i}``, ``_all_layers``, ``input_weights`` / ``hidden_weights``), ``init_hidden``,
``param_count``, batch-first ``forward(input[B,T,in], init_states=None) -> (outputs[B,T,H], (h, c))``
where one ``(h, c)`` pair seeds every layer and the last layer's final state is returned
(lstm.py:117-135).  Gate order i, f, g, o (lstm.py:26-29); the dense cell has a bias on the hidden
weights only (lstm.py:18,21).

What differs: the arithmetic.  ``LSTM.forward`` hands each layer's whole sequence to the
persistent HIP kernel (see ``_fused.py``) instead of looping over timesteps in Python.
"""
import torch
from torch import nn

from ._fused import FusedCellMixin, FusedRnnBase


class LSTMCell(FusedCellMixin, nn.Module):
    kind = 'lstm'
    n_gate = 4

    def __init__(self, input_size, hidden_size, bias, device):
        nn.Module.__init__(self)
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bias = bias
        self.device = device
        # creation order fixes the RNG draw order: input weights first
        self.input_weights = self._create_input_hidden_weights()
        self.hidden_weights = self._create_hidden_hidden_weights()

    def _create_input_hidden_weights(self):
        return nn.Linear(self.input_size, 4 * self.hidden_size, False).to(self.device)

    def _create_hidden_hidden_weights(self):
        return nn.Linear(self.hidden_size, 4 * self.hidden_size, self.bias).to(self.device)

    def forward(self, input, hx, cx):
        """One timestep: (x[B,in], h[B,H], c[B,H]) -> (h', c')."""
        if self._fusable():
            hy, cy = self._fused_step(input, hx, cx)
        else:
            # cells whose weights are neither nn.Linear, TTLinear nor a TTLinearSet the library takes as one joint matrix
            # (_fused.py: _fusable): the weights' own forward + device-side gate arithmetic
            H = self.hidden_size
            pre = self.input_weights(input) + self.hidden_weights(hx)
            i, f, o = (torch.sigmoid(pre[:, k * H:(k + 1) * H]) for k in (0, 1, 3))
            g = torch.tanh(pre[:, 2 * H:3 * H])
            cy = f * cx + i * g
            hy = o * torch.tanh(cy)
        if hasattr(self, '_h_backward_hook') and hy.requires_grad:
            assert hasattr(self, '_c_backward_hook') and cy.requires_grad
            hy.register_hook(self._h_backward_hook)
            cy.register_hook(self._c_backward_hook)
        return hy, cy


class LSTM(FusedRnnBase):
    kind = 'lstm'

    def __init__(self, input_size, hidden_size, num_layers, device, bias=True, log_grads=False):
        super(LSTM, self).__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bias = bias
        self.device = device
        self.log_grads = log_grads
        self._build_layers(log_grads)

    def _create_first_layer_cell(self):
        return LSTMCell(self.input_size, self.hidden_size, self.bias, self.device)

    def _create_other_layer_cell(self):
        return LSTMCell(self.hidden_size, self.hidden_size, self.bias, self.device)

    def init_hidden(self, batch_size):
        h = torch.zeros(batch_size, self.hidden_size).to(self.device)
        c = torch.zeros(batch_size, self.hidden_size).to(self.device)
        return h, c

    def forward(self, input, init_states=None, need_outputs=True):
        """
        :param input:       (batch_size, seq_len, input_size)
        :param init_states: optional (h, c), each (batch_size, hidden_size); seeds every layer.
        :param need_outputs: extension to the reference's signature.  False under torch.no_grad() tells the last layer that
                 only the final state is consumed (speaker_encoder.py:80-86): `outputs` is then None where the kernel can skip
                 the [B, T, H] store, and the usual tensor everywhere else (and always when autograd is recording).
        :return: outputs (batch_size, seq_len, hidden_size) of the last layer, and that layer's
                 final (h, c), each (batch_size, hidden_size).
        """
        if self._needs_stepping():
            h, c = self.init_hidden(input.shape[0]) if init_states is None else init_states
            h, c = h.to(input.dtype), c.to(input.dtype)
            outputs, hT, cT = self._forward_stepwise(input, h, c)
        else:
            h, c = (None, None) if init_states is None else init_states
            outputs, hT, cT = self._forward_fused(input, h, c, need_outputs)
        return outputs, (hT, cT)
