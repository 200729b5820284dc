"""TT-matrix operations of the drop-in ``t3nsor`` API, executed by libttrnn.so on the MI355X.

``tt_dense_matmul`` replaces the reference's einsum chain (``t3nsor/ops.py:54-93``): same
arguments, same result layout, same ``ValueError`` on mismatched inner dimensions — but the whole
core chain runs as one HIP kernel (no ``.contiguous()`` copies, no per-core dispatch).
``transpose`` (``ops.py:47-51``) stays a zero-copy view swap because the parameter stride layout it
produces is part of the checkpoint contract.
"""
from .tensor_train import TensorTrain


def transpose(tt_matrix):
    """Swap the row / column mode axes of every core (views, no copy)."""
    return TensorTrain([core.transpose(1, 2) for core in tt_matrix.tt_cores], convert_to_tensors=False)


def tt_dense_matmul(tt_matrix_a, matrix_b):
    """(TT-matrix M x N) @ (dense N x P) -> dense M x P, on the device."""
    from ttrnn_hip import functional as F
    a_shape = tt_matrix_a.shape
    if a_shape[1] is not None and matrix_b.shape[0] is not None and a_shape[1] != matrix_b.shape[0]:
        raise ValueError('Arguments shapes should align got {} and {} instead.'.format(
            list(a_shape), list(matrix_b.shape)))
    # rows of x = columns of B; the kernel consumes batch-major rows
    y = F.tt_linear(matrix_b.transpose(0, 1), tt_matrix_a.tt_cores, bias=None)
    return y.transpose(0, 1)
