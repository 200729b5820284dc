"""``TTLinearSet`` — the "naive" tensorisation: one independent TTLinear per gate, outputs
concatenated column-wise (reference: ``tensorized_rnn/tt_linearset.py:5-38``).

state_dict keeps the reference's duplicate registration (``gate{i}.*`` attributes AND the
``gates`` ModuleList, tt_linearset.py:23,25).  Each gate runs the fused HIP chain kernel.
"""
import torch
import torch.nn as nn

from t3nsor.layers import TTLinear


class TTLinearSet(nn.Module):
    def __init__(self, in_features=None, out_features=None, n_gates=4, bias=True, init=None, shape=None,
                 auto_shapes=True, d=3, tt_rank=8, auto_shape_mode='ascending',
                 auto_shape_criterion='entropy'):
        super(TTLinearSet, self).__init__()
        self.n_gates = n_gates
        self.in_features = in_features
        self.out_features = out_features
        members = []
        for g in range(n_gates):
            lin = TTLinear(in_features=in_features, out_features=out_features, bias=bias, init=init,
                           shape=shape, auto_shapes=auto_shapes, d=d, tt_rank=tt_rank,
                           auto_shape_mode=auto_shape_mode, auto_shape_criterion=auto_shape_criterion)
            setattr(self, 'gate{}'.format(g), lin)
            members.append(lin)
        self.gates = nn.ModuleList(members)

    def forward(self, x):
        assert x.size(1) == self.in_features
        return torch.cat([gate(x) for gate in self.gates], dim=1)
