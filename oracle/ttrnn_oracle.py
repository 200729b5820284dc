"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's TT-LSTM / TT-GRU path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (tensorized-rnn_amd/) never does and has no CPU path at all.

Parity status: PINNED.  The reference publishes no golden vectors for this path (it has no tests),
so the oracle is pinned against outputs of the reference itself, generated in the build container
by tests/golden/gen_golden.py (which imports /root/reference) and committed as tests/golden/*.npz;
tests/test_oracle_golden.py checks every fixture (forward <= 1e-6 abs, gradients <= 1e-5 of max).

The functions below restate, op for op and in the reference's fp32 arithmetic, what the reference
executes on CPU (file:line under the reference repo).  They take raw tensors (lists of cores with
logical shape (R_k, I_k, J_k, R_{k+1})), not modules, so they share no code with the product.
Because they issue the same ATen op sequence (einsum -> bmm, .contiguous() copies, slice /
sigmoid / tanh, Python time loop) they are also the CPU baseline timed by bench.py
(``cpu_baseline.kind = "port"``).
"""
import torch


def tt_dense_matmul(cores, matrix_b):
    """(TT-matrix M x N given by `cores`) @ (dense N x P).   t3nsor/ops.py:54-93
    cores[k]: (R_k, I_k, J_k, R_{k+1});  matrix_b: (N, P) with N = prod J_k."""
    d = len(cores)
    in_modes = [c.shape[2] for c in cores]
    ranks = [c.shape[0] for c in cores] + [1]
    rows = 1
    for c in cores:
        rows *= c.shape[1]
    # (P, j_0..j_{d-2}) x j_{d-1} x 1                                           ops.py:78-79
    data = matrix_b.transpose(0, 1).contiguous().view(-1, in_modes[-1], 1)
    for k in reversed(range(d)):
        # (a i j b), (r j b) -> (i r a): contract mode j_k and rank b              ops.py:85
        data = torch.einsum('aijb,rjb->ira', [cores[k], data])
        if k > 0:
            # regroup as (i_k.., P, j_0..j_{k-2}) x j_{k-1} x R_k                  ops.py:89-90
            data = data.contiguous().view(-1, in_modes[k - 1], ranks[k])
    return data.view(rows, matrix_b.shape[1])                                   # ops.py:93


def ttlinear(cores, bias, x):
    """TTLinear.forward: x[B,in] -> y[B,out].                        t3nsor/layers.py:121-127"""
    y = tt_dense_matmul(cores, x.transpose(0, 1)).transpose(0, 1)
    return y if bias is None else y + bias


def linear_set(gates, x):
    """TTLinearSet.forward: per-gate TTLinears concatenated.   tensorized_rnn/tt_linearset.py:27-38
    gates: list of (cores, bias)."""
    return torch.cat([ttlinear(c, b, x) for (c, b) in gates], dim=1)


def apply_weights(w, x):
    """w is one of: ('tt', cores, bias) | ('set', [(cores, bias), ...]) | ('dense', weight, bias)."""
    if w[0] == 'tt':
        return ttlinear(w[1], w[2], x)
    if w[0] == 'set':
        return linear_set(w[1], x)
    y = x @ w[1].t()
    return y if w[2] is None else y + w[2]


def lstm_cell(w_in, w_hid, x, hx, cx):
    """LSTMCell.forward, gate order i,f,g,o.                      tensorized_rnn/lstm.py:23-32"""
    H = hx.shape[1]
    gates = apply_weights(w_in, x) + apply_weights(w_hid, hx)
    i = torch.sigmoid(gates[:, :H])
    f = torch.sigmoid(gates[:, H:2 * H])
    g = torch.tanh(gates[:, 2 * H:3 * H])
    o = torch.sigmoid(gates[:, 3 * H:])
    cy = (f * cx) + (i * g)
    hy = o * torch.tanh(cy)
    return hy, cy


def gru_cell(w_in, w_hid, x, hx):
    """GRUCell.forward, gate order r,z,n.                          tensorized_rnn/gru.py:25-44"""
    H = hx.shape[1]
    gi = apply_weights(w_in, x)
    gh = apply_weights(w_hid, hx)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * hx


def lstm_forward(layers, x, init_states=None):
    """LSTM.forward: step-major loop over layers; one (h, c) seeds all layers; returns the last
    layer's outputs and final state.                            tensorized_rnn/lstm.py:101-135
    layers: list of (w_in, w_hid)."""
    B, T, _ = x.shape
    H = _hidden_size(layers[0][1])
    outputs = torch.zeros(B, T, H, dtype=x.dtype)
    if init_states is None:
        h, c = torch.zeros(B, H, dtype=x.dtype), torch.zeros(B, H, dtype=x.dtype)
    else:
        h, c = init_states
    state = [(h, c)] * len(layers)
    inp = new_c = None
    for t in range(T):
        inp = x[:, t, :]
        for l, (w_in, w_hid) in enumerate(layers):
            h, c = state[l]
            inp, new_c = lstm_cell(w_in, w_hid, inp, h, c)
            state[l] = (inp, new_c)
        outputs[:, t, :] = inp
    return outputs, (inp, new_c)


def gru_forward(layers, x, init_states=None):
    """GRU.forward.                                               tensorized_rnn/gru.py:104-136"""
    B, T, _ = x.shape
    H = _hidden_size(layers[0][1])
    outputs = torch.zeros(B, T, H, dtype=x.dtype)
    h = torch.zeros(B, H, dtype=x.dtype) if init_states is None else init_states
    state = [h] * len(layers)
    inp = None
    for t in range(T):
        inp = x[:, t, :]
        for l, (w_in, w_hid) in enumerate(layers):
            inp = gru_cell(w_in, w_hid, inp, state[l])
            state[l] = inp
        outputs[:, t, :] = inp
    return outputs, inp


def _hidden_size(w_hid):
    if w_hid[0] == 'tt':
        n = 1
        for c in w_hid[1]:
            n *= c.shape[2]
        return n
    if w_hid[0] == 'set':
        n = 1
        for c in w_hid[1][0][0]:
            n *= c.shape[2]
        return n
    return w_hid[1].shape[1]


# --------------------------------------------------------------------------------------------------
# helpers to rebuild the oracle's operands from a golden case / from shapes
# --------------------------------------------------------------------------------------------------
def layers_from_state_dict(sd, num_layers, requires_grad=False, dtype=torch.float32):
    """sd: {key: tensor} with the reference's state_dict keys.  Returns the
    `layers` list for lstm_forward / gru_forward plus a {key: tensor} dict of the leaf tensors.
    dtype=torch.float64 evaluates the same op sequence in double (the "exact" result for error measurements)."""
    leaves = {}

    def leaf(key):
        t = sd[key].clone().to(dtype)
        if requires_grad:
            t.requires_grad_(True)
        leaves[key] = t
        return t

    def weights(prefix):
        if prefix + 'weight' in sd:                      # dense nn.Linear
            b = leaf(prefix + 'bias') if prefix + 'bias' in sd else None
            return ('dense', leaf(prefix + 'weight'), b)
        if prefix + 'gates.0.parameters.0' in sd:        # TTLinearSet (gate{i} duplicates gates.{i})
            gates = []
            g = 0
            while prefix + 'gates.%d.parameters.0' % g in sd:
                gp = prefix + 'gates.%d.' % g
                gates.append(_tt(gp))
                g += 1
            return ('set', [(c, b) for (_, c, b) in gates])
        return _tt(prefix)

    def _tt(prefix):
        cores = []
        k = 0
        while prefix + 'parameters.%d' % k in sd:
            cores.append(leaf(prefix + 'parameters.%d' % k))
            k += 1
        b = leaf(prefix + 'bias') if prefix + 'bias' in sd else None
        return ('tt', cores, b)

    layers = []
    for l in range(num_layers):
        layers.append((weights('cell%d.input_weights.' % l), weights('cell%d.hidden_weights.' % l)))
    return layers, leaves


def random_tt(in_modes, out_modes, rank, generator=None, bias=True):
    """Glorot-scaled random TT weights of the given mode shapes (for timing / property tests)."""
    d = len(in_modes)
    ranks = [1] + [rank] * (d - 1) + [1]
    n_in = 1
    n_out = 1
    for j, i in zip(in_modes, out_modes):
        n_in *= j
        n_out *= i
    lamb = 2.0 / (n_in + n_out)
    std = (lamb ** 0.5) ** (1.0 / d)
    for r in ranks:
        std *= r ** (-1.0 / (2 * d))
    cores = [torch.randn(ranks[k], out_modes[k], in_modes[k], ranks[k + 1], generator=generator) * std
             for k in range(d)]
    b = 1e-3 * torch.ones(n_out) if bias else None
    return ('tt', cores, b)
