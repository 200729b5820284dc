"""TensorTrain container — the weight layout contract of the drop-in API.

Same public surface as the reference's ``t3nsor/tensor_train.py:6-163`` (tt_cores, raw_shape,
shape, ranks, ndims, is_tt_matrix, dof, total, to / detach / requires_grad_ / to_parameter / full),
re-implemented.  Only what the TT-LSTM / TT-GRU path touches is provided (the reference's
``TensorTrainBatch`` is used solely by dead code and is out of scope).

Layout facts the kernels rely on (probed on the reference):
  * a TT-matrix core is 4-d ``(R_k, n_k, m_k, R_{k+1})``; a TT-tensor core is 3-d;
  * ``to_parameter`` wraps every core — including non-contiguous transposed views — in an
    ``nn.Parameter`` tagged ``is_tt = True`` and collects them in an ``nn.ParameterList``, so the
    strides of the original storage survive ``state_dict`` round trips.
"""
import numpy as np
import torch
import torch.nn as nn


class TensorTrain(object):
    def __init__(self, tt_cores, shape=None, tt_ranks=None, convert_to_tensors=True):
        cores = list(tt_cores)
        if convert_to_tensors:
            cores = [c if torch.is_tensor(c) else torch.Tensor(c) for c in cores]
        self._tt_cores = cores
        self._is_tt_matrix = cores[0].dim() == 4
        if self._is_tt_matrix:
            self._raw_shape = [[int(c.shape[1]) for c in cores], [int(c.shape[2]) for c in cores]]
            self._shape = [int(np.prod(self._raw_shape[0])), int(np.prod(self._raw_shape[1]))]
            self._ndims = len(cores)
        else:
            self._raw_shape = [int(c.shape[1]) for c in cores]
            self._shape = list(self._raw_shape)
            self._ndims = len(cores)
        self._ranks = [int(c.shape[0]) for c in cores] + [1]
        self._is_parameter = False
        self._parameter = None
        self._dof = int(sum(int(np.prod(list(c.shape))) for c in cores))
        self._total = int(np.prod(self._shape))

    tt_cores = property(lambda self: self._tt_cores)
    raw_shape = property(lambda self: self._raw_shape)
    is_tt_matrix = property(lambda self: self._is_tt_matrix)
    shape = property(lambda self: self._shape)
    ranks = property(lambda self: self._ranks)
    ndims = property(lambda self: self._ndims)
    is_parameter = property(lambda self: self._is_parameter)
    dof = property(lambda self: self._dof)
    total = property(lambda self: self._total)

    @property
    def parameter(self):
        if not self._is_parameter:
            raise ValueError('Not a parameter, run .to_parameter() first')
        return self._parameter

    def _map(self, fn):
        return TensorTrain([fn(c) for c in self._tt_cores], convert_to_tensors=False)

    def to(self, device):
        return self._map(lambda c: c.to(device))

    def detach(self):
        return self._map(lambda c: c.detach())

    def requires_grad_(self, requires_grad=True):
        return self._map(lambda c: c.requires_grad_(requires_grad))

    def to_parameter(self):
        params = []
        for core in self._tt_cores:
            p = nn.Parameter(core)
            p.is_tt = True
            params.append(p)
        out = TensorTrain(params, convert_to_tensors=False)
        out._parameter = nn.ParameterList(out.tt_cores)
        out._is_parameter = True
        return out

    def full(self):
        """Dense tensor / matrix.  Unlike the reference (tensor_train.py:124, which raises on the
        transposed non-contiguous cores of a TTLinear weight) this works for any strides."""
        res = self._tt_cores[0]
        for k in range(1, self._ndims):
            res = res.reshape(-1, self._ranks[k]) @ self._tt_cores[k].reshape(self._ranks[k], -1)
        if not self._is_tt_matrix:
            return res.reshape(*self._shape)
        inter = []
        for n, m in zip(*self._raw_shape):
            inter += [n, m]
        d = self._ndims
        res = res.reshape(*inter).permute(*(list(range(0, 2 * d, 2)) + list(range(1, 2 * d, 2))))
        return res.reshape(*self._shape)

    def __str__(self):
        dev = self._tt_cores[0].device
        rate = self._total / float(self._dof)
        if self._is_tt_matrix:
            return ("A TT-Matrix of size %d x %d, underlying tensorshape: %s x %s, TT-ranks: %s \n"
                    " on device '%s' with compression rate %.2f" % (
                        self._shape[0], self._shape[1], self._raw_shape[0], self._raw_shape[1],
                        self._ranks, dev, rate))
        return ("A Tensor Train of shape %s, TT-ranks: %s\n on device '%s' with compression rate %.2f"
                % (self._shape, self._ranks, dev, rate))
