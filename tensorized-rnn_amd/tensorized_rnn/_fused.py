"""Shared machinery of the drop-in LSTM / GRU modules.

The reference runs ``for step in range(seq_len): for layer in layers: cell(...)`` in Python
(``tensorized_rnn/lstm.py:123-133``, ``gru.py:124-134``).  Here every layer is ONE call into
libttrnn.so that keeps the sequence loop on the GPU (``ttrnn_rnn_forward``); layer l+1 consumes the
full output sequence of layer l, which is the same computation re-ordered (layer l at step t only
depends on layer l-1 at step t and on itself at step t-1).

Dense (``nn.Linear``) cells ride the same kernels: a dense matrix is a TT-matrix with a single
core ``(1, out, in, 1)``.

``log_grads=True`` stays on the fused path: the per-step statistics ``ActivGradLogger`` records through
forward / tensor hooks in the reference come out of the sequence call itself (``ttrnn_hip.functional.StepStats``:
activations from ``out`` and the saved ``c_t``, gradients from the ``d_state`` output of the reverse-time kernel).

``is_naive=True`` cells (``TTLinearSet``: one TT-matrix per gate) take the same fused path: the set is handed to
the library as one TT-matrix with a leading gate-selector core (``TTLinearSet.joint_cores``).  Stepping the cells one
timestep at a time (``_forward_stepwise``) remains only for cells whose weights are neither ``nn.Linear``, ``TTLinear``
nor ``TTLinearSet``.
"""
import torch
from torch import nn


def _lin_as_tt(linear):
    """nn.Linear weight [out, in] viewed as the single core (1, out, in, 1) of a d=1 TT-matrix."""
    w = linear.weight
    return [w.view(1, w.shape[0], w.shape[1], 1)], linear.bias


class FusedCellMixin(object):
    """Adds `_operands()` / `_layer_spec()` to a cell that owns input_weights / hidden_weights."""
    kind = None      # 'lstm' | 'gru'

    def _fusable(self):
        """Can both weights be handed to the library as TT-matrices?  Decided from the weight TYPES alone (no tensors are
        built): nn.Linear (one core), TTLinear, or a TTLinearSet whose joint matrix (one core more than a gate's) still
        fits the library's core limit."""
        from ttrnn_hip._lib import TTRNN_MAX_D
        for w in (self.input_weights, self.hidden_weights):
            if isinstance(w, nn.Linear) or hasattr(w, 'weight_t'):
                continue
            if hasattr(w, 'joint_cores') and len(w.gates[0].weight_t.tt_cores) + 1 <= TTRNN_MAX_D:
                continue
            return False
        return True

    def _operands(self):
        """(cores_in, bias_in, cores_hid, bias_hid), or None when the cell cannot be expressed as two TT matrices the
        library takes (`_fusable`): such cells are stepped."""
        if not self._fusable():
            return None
        ops = []
        for w in (self.input_weights, self.hidden_weights):
            if isinstance(w, nn.Linear):
                ops.append(_lin_as_tt(w))
            elif hasattr(w, 'weight_t'):
                ops.append((list(w.weight_t.tt_cores), w.bias))
            elif hasattr(w, 'joint_cores'):
                ops.append(w.joint_cores())          # naive per-gate set as ONE TT-matrix (tt_linearset.py)
            else:
                return None
        return ops[0][0], ops[0][1], ops[1][0], ops[1][1]

    def _layer_spec(self):
        spec = getattr(self, '_spec_cache', None)
        if spec is None:
            from ttrnn_hip.functional import RnnLayerSpec, TTSpec
            cin, bin_, chid, bhid = self._operands()
            blocks = self.hidden_weights.n_gates if hasattr(self.hidden_weights, 'joint_cores') else 1
            spec = RnnLayerSpec(self.kind, self.input_size, self.hidden_size, TTSpec.from_cores(cin),
                                TTSpec.from_cores(chid), bin_ is not None, bhid is not None, hid_blocks=blocks)
            # plain attribute (bypasses nn.Module bookkeeping); rebuilt lazily after unpickling
            object.__setattr__(self, '_spec_cache', spec)
        return spec

    def _fused_step(self, x, hx, cx=None):
        """One timestep through the fused kernel (T = 1)."""
        from ttrnn_hip import functional as F
        cin, bin_, chid, bhid = self._operands()
        res = F.tt_rnn_layer(self._layer_spec(), x.unsqueeze(1), hx, cx, cin, bin_, chid, bhid)
        return res[1:]          # (hy, cy) or (hy,)

    def _run_sequence(self, seq, h0, c0=None, need_out=True, x_bounded=False):
        from ttrnn_hip import functional as F
        cin, bin_, chid, bhid = self._operands()
        stats = None
        if getattr(self, '_h_logger', None) is not None:       # log_grads=True: per-step statistics from the fused call
            stats = F.StepStats(self._h_logger, getattr(self, '_c_logger', None))
        prep = getattr(self, '_prepared', None)
        if prep is not None and not torch.is_grad_enabled():
            if not prep.fresh():                                # parameters were updated in place since: prepare again
                prep = self._prepare()
        else:
            prep = None
        return F.tt_rnn_layer(self._layer_spec(), seq, h0, c0, cin, bin_, chid, bhid, stats=stats, prepared=prep,
                              need_out=need_out, x_bounded=x_bounded)

    def _prepare(self):
        """(Re)build the weight-only state of this cell for no-grad forwards (ttrnn_hip.functional.PreparedLayer)."""
        from ttrnn_hip import functional as F
        cin, bin_, chid, bhid = self._operands()
        # the freshness stamp watches the cell's own Parameters (a naive set's joint cores are derived copies)
        params = list(self.input_weights.parameters()) + list(self.hidden_weights.parameters())
        prep = F.PreparedLayer(self._layer_spec(), cin, bin_, chid, bhid, params=params)
        object.__setattr__(self, '_prepared', prep)            # plain attribute: not a buffer, not in the state_dict
        return prep

    def _release_prepared(self):
        object.__setattr__(self, '_prepared', None)

    def __getstate__(self):
        # pickling (torch.save(model)) and copy.deepcopy go through here: the prepared state (packed cores, device
        # workspaces) and the descriptor cache (ctypes structures) belong to THIS module on THIS device and are rebuilt on demand
        state = dict(self.__dict__)
        state.pop('_prepared', None)
        state.pop('_spec_cache', None)
        return state


class TTWeightsMixin(object):
    """TT weight factories shared by ``TTLSTMCell`` and ``TTGRUCell``.

    Reference behaviour being mirrored (``tt_lstm.py:13-40``, ``gru.py:139-163``): both weight sets are a
    gates-concatenated ``TTLinear`` (``n_gate * hidden`` outputs, mode shapes from ``tt_shape``) or, for
    ``is_naive``, a ``TTLinearSet`` of ``n_gate`` independent TTLinears.  ``naive_bias`` is what the
    naive set receives for ``bias``: the LSTM passes a literal False (tt_lstm.py:21,34), the GRU passes
    the cell's flag (gru.py:150,163)."""
    n_gate = None
    naive_bias = None       # None -> follow self.bias

    def _tt_options(self, n_cores, tt_rank, is_naive, new_core):
        # must run BEFORE the dense base __init__, which calls the factories below
        self.n_cores, self.tt_rank = n_cores, tt_rank
        self.is_naive, self.new_core = is_naive, new_core

    def _tt_weights(self, fan_in):
        from t3nsor.layers import TTLinear
        from .rnn_utils import tt_shape
        from .tt_linearset import TTLinearSet
        G, H = self.n_gate, self.hidden_size
        if self.is_naive:
            b = self.bias if self.naive_bias is None else self.naive_bias
            w = TTLinearSet(in_features=fan_in, out_features=H, n_gates=G, bias=b, auto_shapes=True,
                            d=self.n_cores, tt_rank=self.tt_rank)
        else:
            modes = tt_shape(fan_in, H, self.n_cores, G, new_core=self.new_core)
            w = TTLinear(out_features=G * H, shape=modes, bias=self.bias, auto_shapes=False,
                         d=self.n_cores, tt_rank=self.tt_rank)
        return w.to(self.device)

    def _create_input_hidden_weights(self):
        return self._tt_weights(self.input_size)

    def _create_hidden_hidden_weights(self):
        return self._tt_weights(self.hidden_size)


class TTStackMixin(object):
    """Layer factories of the multi-layer TT models: every layer gets a cell of ``tt_cell_cls``."""
    tt_cell_cls = None

    def _tt_options(self, n_cores, tt_rank, is_naive, new_core):
        assert new_core in [None, 'first', 'last']
        self.n_cores, self.tt_rank = n_cores, tt_rank
        self.is_naive, self.new_core = is_naive, new_core

    def _tt_cell(self, fan_in):
        return self.tt_cell_cls(fan_in, self.hidden_size, self.bias, self.device, n_cores=self.n_cores,
                                tt_rank=self.tt_rank, is_naive=self.is_naive, new_core=self.new_core)

    def _create_first_layer_cell(self):
        return self._tt_cell(self.input_size)

    def _create_other_layer_cell(self):
        return self._tt_cell(self.hidden_size)


class FusedRnnBase(nn.Module):
    """Layer stack + state handling shared by LSTM and GRU."""
    kind = None

    def _build_layers(self, log_grads):
        from .rnn_utils import ActivGradLogger
        self._all_layers = []
        for i in range(self.num_layers):
            cell = self._create_first_layer_cell() if i == 0 else self._create_other_layer_cell()
            setattr(self, 'cell{}'.format(i), cell)
            self._all_layers.append(cell)
        if log_grads:
            for i, cell in enumerate(self._all_layers):
                h_logger = ActivGradLogger("hidden_{}".format(i))
                h_fwd, h_bwd = h_logger.create_hooks(0)
                cell.register_forward_hook(h_fwd)
                cell._h_backward_hook = h_bwd
                object.__setattr__(cell, '_h_logger', h_logger)       # fused path: fed by ttrnn_hip.functional.StepStats
                if self.kind == 'lstm':
                    c_logger = ActivGradLogger("cell_{}".format(i))
                    c_fwd, c_bwd = c_logger.create_hooks(1)
                    cell.register_forward_hook(c_fwd)
                    cell._c_backward_hook = c_bwd
                    object.__setattr__(cell, '_c_logger', c_logger)

    def prepare_for_inference(self):
        """Opt-in for repeated no-grad forwards on unchanged weights (the reference's eval loop, experiments/
        digit_classification/benchmarking.py:16-38): every layer keeps its packed cores and, per input shape, the
        weight-only part of the forward call, so that a forward is one recurrent-kernel launch per layer
        (include/ttrnn.h: ttrnn_rnn_forward_phase).  In-place parameter updates are detected and prepared again; writes
        through `.data` are not — call this again (or release_prepared()) after them.  Moving the module (.to / .cuda /
        dtype casts) and train() drop the prepared state.  Returns self."""
        if not self._needs_stepping():
            for cell in self._all_layers:
                cell._prepare()
        return self

    def release_prepared(self):
        for cell in getattr(self, '_all_layers', ()):
            cell._release_prepared()
        return self

    def train(self, mode=True):
        if mode:
            self.release_prepared()
        return super(FusedRnnBase, self).train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.release_prepared()
        return super(FusedRnnBase, self)._apply(fn, *args, **kwargs)

    def param_count(self):
        from .rnn_utils import param_count as pc
        return sum(pc(getattr(cell, attr)) for cell in self._all_layers
                   for attr in ('input_weights', 'hidden_weights'))

    def _needs_stepping(self):
        # log_grads=True and the naive per-gate cells (TTLinearSet as one joint TT-matrix) stay on the fused path; only
        # cells with foreign weight types, or a joint matrix with more cores than the library takes, are stepped
        return any(not cell._fusable() for cell in self._all_layers)

    def _forward_fused(self, input, h0, c0, need_outputs=True):
        seq = input
        last = None
        n = len(self._all_layers)
        for i, cell in enumerate(self._all_layers):
            # only the LAST layer's outputs can go unused (layer l + 1 reads all of layer l's)
            # a stacked layer's input is the hidden-state sequence below: |x| <= 1 (LSTM: o * tanh(c); GRU: convex combinations
            # of tanh values and h_{t-1}, from |h_0| <= 1 on) — the weight-gradient step then skips its column-maximum pass
            bounded = i > 0 and (self.kind == 'lstm' or h0 is None)
            last = cell._run_sequence(seq, h0, c0, need_out=need_outputs or i + 1 < n, x_bounded=bounded)
            seq = last[0]
        return last

    def _forward_stepwise(self, input, h0, c0):
        """Step-major loop with per-step cell calls (hooks fire per step, as in the reference)."""
        lstm = self.kind == 'lstm'
        state = [(h0, c0)] * self.num_layers
        steps = []
        x = None
        for t in range(input.shape[1]):
            x = input[:, t, :]
            for i, cell in enumerate(self._all_layers):
                h, c = state[i]
                if lstm:
                    x, c = cell(x, h, c)
                else:
                    x = cell(x, h)
                state[i] = (x, c)
            steps.append(x)
        outputs = torch.stack(steps, dim=1)
        return outputs, state[-1][0], state[-1][1]
