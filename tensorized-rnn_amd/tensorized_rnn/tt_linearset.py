"""``TTLinearSet`` — the "naive" tensorisation: one independent TTLinear per gate, outputs
concatenated column-wise (reference: ``tensorized_rnn/tt_linearset.py:5-38``).

state_dict keeps the reference's duplicate registration (``gate{i}.*`` attributes AND the
``gates`` ModuleList, tt_linearset.py:23,25).  Each gate runs the fused HIP chain kernel.
"""
import torch
from torch import nn

from t3nsor.layers import TTLinear


class TTLinearSet(nn.Module):
    def __init__(self, in_features=None, out_features=None, n_gates=4, bias=True, init=None, shape=None,
                 auto_shapes=True, d=3, tt_rank=8, auto_shape_mode='ascending',
                 auto_shape_criterion='entropy'):
        nn.Module.__init__(self)
        self.in_features, self.out_features, self.n_gates = in_features, out_features, n_gates
        per_gate = dict(in_features=in_features, out_features=out_features, bias=bias, init=init,
                        shape=shape, auto_shapes=auto_shapes, d=d, tt_rank=tt_rank,
                        auto_shape_mode=auto_shape_mode, auto_shape_criterion=auto_shape_criterion)
        # RNG draw order = gate order; each gate is registered twice (attribute first, list second),
        # which is what fixes the key order of the reference's state_dict.
        built = []
        for index in range(n_gates):
            built.append(TTLinear(**per_gate))
            self.add_module('gate%d' % index, built[-1])
        self.gates = nn.ModuleList(built)

    def forward(self, x):
        if x.size(1) != self.in_features:
            raise AssertionError('TTLinearSet: expected %d input features, got %d'
                                 % (self.in_features, x.size(1)))
        return torch.cat([member(x) for member in self.gates], dim=1)
