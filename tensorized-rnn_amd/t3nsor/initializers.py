"""TT-matrix initialisers used by TTLinear (init parity with the reference).

Restates ``t3nsor/initializers.py:166-215`` (matrix_with_random_cores), ``:218-282``
(random_matrix) and ``:286-300`` (glorot_initializer) of the reference: same scalar arithmetic
(numpy float64), same ``torch.randn`` call sequence (core 0 .. d-1, each of shape
``(R_k, n_k, m_k, R_{k+1})``), so the same seed gives bit-identical cores
(golden: tests/golden/g7_init_*.npz).  The reference's unused zeros / ones / eye / batch variants
are out of scope.
"""
import numpy as np
import torch

from .tensor_train import TensorTrain


def _normalise(shape, tt_rank):
    shape = list(shape)
    if shape[0] is None:
        shape[0] = np.ones(len(shape[1]), dtype=int)
    if shape[1] is None:
        shape[1] = np.ones(len(shape[0]), dtype=int)
    shape = np.array(shape)
    tt_rank = np.array(tt_rank)
    if shape.ndim != 2 or shape.shape[0] != 2:
        raise ValueError('shape should be 2d array, got %a' % (shape,))
    if shape[0].size != shape[1].size:
        raise ValueError('shape[0] should have the same length as shape[1], but'
                         '%d != %d' % (shape[0].size, shape[1].size))
    if np.any(shape.flatten() < 1):
        raise ValueError('all elements in `shape` should be positive, got %a' % (shape,))
    if not all(isinstance(v, (int, np.integer)) for v in shape.flatten()):
        raise ValueError('all elements in `shape` should be integers, got %a' % (shape,))
    if np.any(tt_rank < 1):
        raise ValueError('`rank` should be positive, got %a' % (tt_rank,))
    d = shape[0].size
    if tt_rank.size not in (1, d + 1):
        raise ValueError('`rank` array has inappropriate size, expected 1 or %d, got %d' % (d + 1, tt_rank.size))
    if tt_rank.size == 1:
        tt_rank = np.concatenate([[1], tt_rank * np.ones(d - 1), [1]])
    return shape, tt_rank.astype(int), d


def matrix_with_random_cores(shape, tt_rank=2, mean=0., stddev=1., dtype=torch.float32):
    """TT-matrix whose cores are i.i.d. N(mean, stddev^2)."""
    shape, ranks, d = _normalise(shape, tt_rank)
    cores = []
    for k in range(d):
        dims = (int(ranks[k]), int(shape[0][k]), int(shape[1][k]), int(ranks[k + 1]))
        cores.append(torch.randn(dims, dtype=dtype) * stddev + mean)
    return TensorTrain(cores)


def random_matrix(shape, tt_rank=2, mean=0., stddev=1., dtype=torch.float32):
    """Random TT-matrix whose dense entries have the requested stddev (mean 0 only)."""
    shape, ranks, d = _normalise(shape, tt_rank)
    # entries of a TT product of N(0,1) cores have variance prod(ranks): spread the correction
    # evenly over the d cores
    exponent = -1.0 / (2 * d)
    scale = np.prod(ranks ** exponent)
    core_stddev = stddev ** (1.0 / d) * scale
    tt = matrix_with_random_cores(shape, tt_rank=ranks, stddev=core_stddev, dtype=dtype)
    if np.abs(mean) >= 1e-8:
        raise NotImplementedError('non-zero mean is not supported yet')
    return tt


def glorot_initializer(shape, tt_rank=2, dtype=torch.float32):
    shape_a, _, _ = _normalise(shape, tt_rank)
    n_in = np.prod(shape_a[0])
    n_out = np.prod(shape_a[1])
    lamb = 2.0 / (n_in + n_out)
    return random_matrix(shape_a, tt_rank=tt_rank, stddev=np.sqrt(lamb), dtype=dtype)
