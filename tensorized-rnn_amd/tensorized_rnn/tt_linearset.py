"""``TTLinearSet`` — the "naive" tensorisation: one independent TTLinear per gate, outputs
concatenated column-wise (reference: ``tensorized_rnn/tt_linearset.py:5-38``).

state_dict keeps the reference's duplicate registration (``gate{i}.*`` attributes AND the
``gates`` ModuleList, tt_linearset.py:23,25).  Each gate runs the fused HIP chain kernel.

For the persistent sequence kernels the set is presented as ONE TT-matrix (``joint_cores``): the direct sum of the
per-gate matrices IS a tensor train with one more core — a leading (1 -> n_gates) "gate selector" mode and block-diagonal
cores of rank n_gates * r — whose output index is ``gate * H + i``, exactly the column order of the concatenation
(tt_linearset.py:33-38).  The assembly is a handful of differentiable tensor ops on KB-sized cores, so autograd scatters
the joint cores' gradients back onto the per-gate Parameters and ``is_naive=True`` models run the same fused time loop
(libttrnn's runtime-shape MFMA kernels) as every other TT shape instead of T x L Python cell calls.
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
        # constants of joint_cores: the gate selector's identity and the block mask — non-persistent buffers (not in the
        # state_dict, whose keys stay the reference's), moved by .to() / .cuda() with the parameters and never allocated lazily
        # inside a forward (a first call under hipGraph capture would put them into the graph's pool: ADVICE r5)
        self.register_buffer('_gate_eye', torch.eye(n_gates), persistent=False)
        self.register_buffer('_gate_mask', torch.eye(n_gates, dtype=torch.bool).view(n_gates, 1, 1, 1, n_gates, 1), persistent=False)

    def joint_cores(self):
        """(cores, bias) of the single TT-matrix  x -> cat_g TT_g(x):  d + 1 cores with logical shapes
        (1, G, 1, G), (G, I_0, J_0, G r_1), (G r_1, I_1, J_1, G r_2), ..., (G r_{d-1}, I_{d-1}, J_{d-1}, 1)."""
        G = self.n_gates
        per_gate = [list(member.weight_t.tt_cores) for member in self.gates]
        d = len(per_gate[0])
        ref = per_gate[0][0]
        eye = self._gate_eye
        if eye.dtype != ref.dtype or eye.device != ref.device:      # (parameters converted behind the module's back)
            eye = eye.to(device=ref.device, dtype=ref.dtype)
        mask = self._gate_mask if self._gate_mask.device == ref.device else self._gate_mask.to(ref.device)
        cores = [eye.view(1, G, 1, G)]
        for k in range(d):
            if k == d - 1:
                cores.append(torch.cat([per_gate[g][k] for g in range(G)], dim=0))      # ranks (G r_{d-1}) -> 1: stacked along dim 0
                continue
            # block (g, h) of the joint core = gate g's core where g == h, +0.0 elsewhere: ONE stack and ONE select instead of two
            # zero blocks and a concatenation per gate (the assembly was ~40 small launches per forward).  A select, not a
            # multiplication by the identity: c * 0 is -0.0 for negative entries and NaN for a non-finite one, which the
            # block-promise check of the kernels (TTRNN_STAT_BLOCK_VIOLATIONS) would report exactly when a run diverges
            c = torch.stack([per_gate[g][k] for g in range(G)], dim=0)                   # (G, r_k, I_k, J_k, r_{k+1})
            blk = torch.where(mask, c.unsqueeze(4), c.new_zeros(()))                      # (G, r_k, I_k, J_k, G, r_{k+1})
            cores.append(blk.reshape(G * c.shape[1], c.shape[2], c.shape[3], G * c.shape[4]))
        biases = [member.bias for member in self.gates]
        bias = None if biases[0] is None else torch.cat(biases, dim=0)
        return cores, bias

    def forward(self, x):
        if x.size(1) != self.in_features:
            raise AssertionError('TTLinearSet: expected %d input features, got %d'
                                 % (self.in_features, x.size(1)))
        return torch.cat([member(x) for member in self.gates], dim=1)
