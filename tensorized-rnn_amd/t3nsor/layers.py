"""``TTLinear`` — the tensor-train linear layer of the drop-in API.

Mirrors the constructor, attributes, ``state_dict`` keys and construction-time print of the
reference's ``t3nsor/layers.py:83-127``; ``forward`` runs the fused HIP chain kernel
(``ttrnn_ttlinear_forward``) instead of d einsum + copy dispatches.

Contract notes:
  * ``self.parameters`` is the ``nn.ParameterList`` of the cores (registered as a sub-module under
    that name, so state_dict keys read ``parameters.{k}``); ``nn.Module.parameters()`` keeps working
    because the class method wins normal attribute lookup;
  * the cores are the transposed views of a glorot TT-matrix: logical ``(R_k, I_k, J_k, R_{k+1})``
    over ``(R_k, J_k, I_k, R_{k+1})`` storage;
  * bias is initialised to 1e-3 (layers.py:116).
The reference's ``TTEmbedding`` is never instantiated by any experiment and is out of scope.
"""
import torch
import torch.nn as nn

from . import utils
from .initializers import glorot_initializer
from .ops import transpose


class TTLinear(nn.Module):
    def __init__(self, in_features=None, out_features=None, bias=True, init=None, shape=None,
                 auto_shapes=True, d=3, tt_rank=8, auto_shape_mode='ascending',
                 auto_shape_criterion='entropy'):
        super(TTLinear, self).__init__()
        if auto_shapes:
            if in_features is None or out_features is None:
                raise ValueError("Shape is not specified")
            shape = [utils.auto_shape(in_features, d=d, criterion=auto_shape_criterion, mode=auto_shape_mode),
                     utils.auto_shape(out_features, d=d, criterion=auto_shape_criterion, mode=auto_shape_mode)]
        if init is None:
            if shape is None:
                raise ValueError("if init is not provided, please specify shape, or set auto_shapes=True")
            init = glorot_initializer(shape, tt_rank=tt_rank)
        else:
            shape = init.raw_shape
        self.shape = shape
        self.weight_t = transpose(init).to_parameter()
        self.parameters = self.weight_t.parameter
        if bias:
            self.bias = nn.Parameter(1e-3 * torch.ones(out_features))
        else:
            self.register_parameter('bias', None)
        self._spec = None
        print('Created TTLinear layer with input shape: {}. output shape: {}'.format(shape[0], shape[1]))

    def tt_spec(self):
        """Static TT description (modes, ranks) handed to libttrnn."""
        if self._spec is None:
            from ttrnn_hip.functional import TTSpec
            self._spec = TTSpec.from_cores(self.weight_t.tt_cores)
        return self._spec

    def forward(self, x):
        from ttrnn_hip import functional as F
        return F.tt_linear(x, self.weight_t.tt_cores, self.bias, spec=self.tt_spec())

    def forward_head(self, x, epilogue):
        """Not in the reference: forward + the caller's row-wise epilogue in ONE library call — `epilogue` is
        "log_softmax" (experiments/digit_classification/mnist_classifier.py:55-57) or "relu_l2norm"
        (experiments/speaker_verification/encoder/speaker_encoder.py:86-89)."""
        from ttrnn_hip import functional as F
        return F.tt_linear_head(x, self.weight_t.tt_cores, self.bias, spec=self.tt_spec(), epilogue=epilogue)
