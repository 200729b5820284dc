"""Shape selection, parameter counting and the activation / gradient logger of the drop-in API.

``tt_shape`` and ``param_count`` follow the reference's ``tensorized_rnn/rnn_utils.py:20-36`` and
``:300-309``.  ``ActivGradLogger`` keeps the reference's public surface (``rnn_utils.py:42-226``:
``all_loggers``, ``create_hooks``, ``end_minibatch``, ``end_epoch``, ``get_logs``, ``del_record``)
for ``log_grads=True`` runs.  Modules built with ``log_grads=True`` stay on the fused sequence path: the
recurrent kernels hand the logger its per-step statistics (mean ||h_t||^2, ||c_t||^2 from the forward's
outputs / reserve records, the per-step state gradients from the reverse-time kernel's ``d_state``
output — ``ttrnn_hip.functional.StepStats``, ``tensorized_rnn/_fused.py``), no cell is stepped one
timestep at a time.  The reference's dead ``project_ttgrad`` helper is out of scope.
"""
from collections import deque

import torch

from t3nsor.utils import auto_shape


def tt_shape(in_features, out_features, n_cores, n_gates, new_core=None):
    """[in_modes, out_modes] of the gates-concatenated TT matrix of one cell.

    new_core=None   : factor in_features and n_gates*out_features into n_cores modes each;
    new_core='first': factor out_features alone and prepend a (1 -> n_gates) mode;
    new_core='last' : same, appended."""
    assert new_core in [None, 'first', 'last']
    total_out = out_features * n_gates if new_core is None else out_features
    in_modes = [int(n) for n in auto_shape(in_features, d=n_cores)]
    out_modes = [int(n) for n in auto_shape(total_out, d=n_cores)]
    if new_core == 'first':
        in_modes, out_modes = [1] + in_modes, [n_gates] + out_modes
    elif new_core == 'last':
        in_modes, out_modes = in_modes + [1], out_modes + [n_gates]
    return [in_modes, out_modes]


def param_count(matrix):
    """Number of scalar weights held by a module (dense or TT)."""
    assert isinstance(matrix, torch.nn.Module)
    return sum(int(p.shape.numel()) for p in matrix.parameters())


def av_norm(tensor, average_logs=False):
    """Batch mean of the squared L2 norm (or of its log) over all non-batch axes."""
    norms = (tensor ** 2).sum(list(range(1, tensor.dim())))
    if average_logs:
        norms = torch.log(norms)
    assert norms.dim() == 1
    return norms.mean()


class ActivGradLogger(object):
    """Per-timestep activation / gradient statistics of one state variable (e.g. 'hidden_0')."""
    all_loggers = dict()
    _QUANTITIES = ('act', 'log_act', 'grad', 'log_grad')

    def __init__(self, name):
        assert name not in ActivGradLogger.all_loggers
        ActivGradLogger.all_loggers[name] = self
        self.name = name
        for q in self._QUANTITIES:
            setattr(self, q + '_epoch', [])
            setattr(self, q + '_mini', [])
        self.act, self.log_act = [], []
        self.grad, self.log_grad = deque(), deque()

    # ---- registry-wide helpers ----------------------------------------------------------------
    @staticmethod
    def get_logs():
        """{(variable, quantity): tensor[num_epochs, seq_len]}"""
        return {(var, q): torch.stack(getattr(lg, q + '_epoch'))
                for var, lg in ActivGradLogger.all_loggers.items() for q in ActivGradLogger._QUANTITIES}

    @staticmethod
    def get_logger(name):
        if name in ActivGradLogger.all_loggers:
            return ActivGradLogger.all_loggers[name]
        print("Logger '{}' not yet initialized".format(name))

    @staticmethod
    def end_epoch():
        for lg in ActivGradLogger.all_loggers.values():
            lg._end_epoch()

    @staticmethod
    def end_minibatch():
        for lg in ActivGradLogger.all_loggers.values():
            lg._end_minibatch()

    @staticmethod
    def del_record():
        for lg in ActivGradLogger.all_loggers.values():
            lg._del_record()

    # ---- hooks --------------------------------------------------------------------------------
    def create_hooks(self, output_ind):
        """(forward_hook for the cell module, backward hook for the output tensor)."""
        @torch.no_grad()
        def forward_hook(rnn_cell, inputs, outputs):
            if not isinstance(outputs, tuple):
                assert output_ind == 0
                outputs = (outputs,)
            target = outputs[output_ind].detach()
            self.act.append(av_norm(target))
            self.log_act.append(av_norm(target, average_logs=True))

        @torch.no_grad()
        def backward_hook(grad_out):
            target = grad_out.detach()
            # gradients arrive last timestep first
            self.grad.appendleft(av_norm(target))
            self.log_grad.appendleft(av_norm(target, average_logs=True))

        return forward_hook, backward_hook

    # ---- bookkeeping --------------------------------------------------------------------------
    def _current(self, q):
        v = getattr(self, q)
        return torch.stack(list(v))

    def _end_minibatch(self):
        for q in self._QUANTITIES:
            mini = getattr(self, q + '_mini')
            cur = self._current(q)
            if mini:
                assert len(mini[0]) == len(cur)
            mini.append(cur)
        self._del_record()

    def _end_epoch(self):
        for q in self._QUANTITIES:
            getattr(self, q + '_epoch').append(torch.mean(torch.stack(getattr(self, q + '_mini')), 0))
            setattr(self, q + '_mini', [])

    def _del_record(self):
        del self.act[:]
        del self.log_act[:]
        self.grad.clear()
        self.log_grad.clear()
