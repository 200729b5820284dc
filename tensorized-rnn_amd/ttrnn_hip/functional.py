"""Autograd-aware entry points that route t3nsor.TTLinear and the tensorized_rnn layers through
libttrnn.so.  Every function here needs device tensors on an MI355X; there is no CPU path.

Reference arithmetic replaced (file:line under the reference repo):
  tt_linear      -> t3nsor/layers.py:121-127 + t3nsor/ops.py:54-93
  tt_rnn_layer   -> tensorized_rnn/lstm.py:23-32,123-133 / gru.py:33-44,124-134 for one layer
"""
import collections
import ctypes

import torch

from . import _lib
from ._lib import TTRNN_BF16, TTRNN_F32, TTRNN_GRU, TTRNN_LSTM, RnnDesc, check, make_ttm

_DT = {torch.float32: TTRNN_F32, torch.bfloat16: TTRNN_BF16}

# Optional measurement hook (bench.py): an object with start(name) / stop(name) that records
# events on the current stream around the dominant kernel launches.  None in normal use.
KERNEL_TIMER = None


class _timed(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if KERNEL_TIMER is not None:
            KERNEL_TIMER.start(self.name)

    def __exit__(self, *exc):
        if KERNEL_TIMER is not None:
            KERNEL_TIMER.stop(self.name)
        return False


def _dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.TtrnnError("libttrnn supports float32 and bfloat16 storage, got {}".format(t.dtype))


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.TtrnnError(
                "the TT-RNN hot path runs on the GPU (libttrnn.so, gfx950) only; got a {} tensor. "
                "There is deliberately no CPU fallback.".format(t.device))


def _check_operands(x, named):
    """The kernels read every operand in x's storage dtype on x's device: refuse a mismatch instead of reinterpreting bytes
    (the reference raises torch's own dtype / device errors at this point)."""
    for name, t in named:
        if t is None:
            continue
        if t.device != x.device:
            raise _lib.TtrnnError("{} is on {} but the input is on {}: all operands of a libttrnn call must share one "
                                  "device".format(name, t.device, x.device))
        if t.dtype != x.dtype:
            raise _lib.TtrnnError("{} has dtype {} but the input has dtype {}: libttrnn computes in the input's storage "
                                  "dtype (cast the module or the input)".format(name, t.dtype, x.dtype))


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# Debug switch (tests / tools/stress_determinism.py): fill every buffer this module allocates WITHOUT initialising it —
# outputs, workspaces, packed cores, reserves, gradient buffers — with random bytes (about one fp32 pattern in 256 is a
# NaN or Inf).  A kernel that reads memory nobody wrote, or leaves part of an output unwritten, then changes its results
# from launch to launch instead of depending on what the allocator happened to hand out.
POISON_ALLOCATIONS = False


def _alloc(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON_ALLOCATIONS and t.numel() > 0:
        t.view(-1).view(torch.uint8).copy_(
            torch.randint(0, 256, (t.numel() * t.element_size(),), dtype=torch.uint8, device=device))
    return t


def _workspace(nbytes, device):
    if nbytes == 0:
        return None
    return _alloc(int(nbytes), torch.uint8, device)


class TTSpec(object):
    """Static description of one TT-matrix (modes, ranks) + helpers to pack its cores."""

    def __init__(self, in_modes, out_modes, ranks):
        self.in_modes = [int(v) for v in in_modes]
        self.out_modes = [int(v) for v in out_modes]
        self.ranks = [int(v) for v in ranks]
        self.d = len(self.in_modes)
        self.in_features = 1
        self.out_features = 1
        for j, i in zip(self.in_modes, self.out_modes):
            self.in_features *= j
            self.out_features *= i
        self.desc = make_ttm(self.in_modes, self.out_modes, self.ranks)
        self._packed_elems = None

    @classmethod
    def from_cores(cls, cores):
        # core k has logical shape (R_k, I_k, J_k, R_{k+1})  (t3nsor/ops.py:47-51 after transpose)
        return cls([c.shape[2] for c in cores], [c.shape[1] for c in cores],
                   [c.shape[0] for c in cores] + [cores[-1].shape[3]])

    @property
    def packed_elems(self):
        if self._packed_elems is None:
            n = _lib.load().ttrnn_packed_elems(ctypes.byref(self.desc))
            if n < 0:
                raise _lib.TtrnnError("invalid TT-matrix description {}x{} ranks {}".format(
                    self.in_modes, self.out_modes, self.ranks))
            self._packed_elems = int(n)
        return self._packed_elems

    def _core_args(self, tensors):
        d = self.d
        ptrs = (ctypes.c_void_p * d)(*[t.data_ptr() for t in tensors])
        strides = (ctypes.c_int64 * (4 * d))()
        for k, t in enumerate(tensors):
            exp = (self.ranks[k], self.out_modes[k], self.in_modes[k], self.ranks[k + 1])
            if tuple(t.shape) != exp:
                raise ValueError("core {} has shape {}, expected {}".format(k, tuple(t.shape), exp))
            for q in range(4):
                strides[4 * k + q] = t.stride(q)
        return ptrs, strides

    def pack(self, cores):
        """strided parameter views -> packed fp32 [W_0..W_{d-1} | Wt_0..Wt_{d-1}] on the device."""
        _require_device(*cores)
        lib = _lib.load()
        dt = _dtype_code(cores[0])
        packed = _alloc((self.packed_elems,), torch.float32, cores[0].device)
        ptrs, strides = self._core_args(cores)
        check(lib.ttrnn_pack_cores(ctypes.byref(self.desc), ptrs, strides, dt, _ptr(packed), _stream(packed)),
              "ttrnn_pack_cores")
        return packed

    @staticmethod
    def pack_pair(spec_a, cores_a, spec_b, cores_b):
        """pack() of two TT-matrices (a layer's input and hidden weights) in one launch."""
        _require_device(*(list(cores_a) + list(cores_b)))
        lib = _lib.load()
        dt = _dtype_code(cores_a[0])
        dev = cores_a[0].device
        pa = _alloc((spec_a.packed_elems,), torch.float32, dev)
        pb = _alloc((spec_b.packed_elems,), torch.float32, dev)
        ptrs_a, strides_a = spec_a._core_args(cores_a)
        ptrs_b, strides_b = spec_b._core_args(cores_b)
        check(lib.ttrnn_pack_cores2(ctypes.byref(spec_a.desc), ptrs_a, strides_a, _ptr(pa), ctypes.byref(spec_b.desc),
                                    ptrs_b, strides_b, _ptr(pb), dt, _stream(pa)), "ttrnn_pack_cores2")
        return pa, pb

    def unpack_grads(self, packed_grad, like):
        """packed fp32 gradient -> list of gradient tensors with the layout of `like` (the cores)."""
        lib = _lib.load()
        grads = [torch.empty_strided(c.shape, c.stride(), dtype=c.dtype, device=c.device) for c in like]
        if POISON_ALLOCATIONS:
            for g in grads:
                g.copy_(torch.full_like(g, float("nan")))
        ptrs, strides = self._core_args(grads)
        check(lib.ttrnn_unpack_core_grads(ctypes.byref(self.desc), _ptr(packed_grad), ptrs, strides,
                                          _dtype_code(like[0]), _stream(packed_grad)), "ttrnn_unpack_core_grads")
        return grads


# Use the by-products of the reverse-time kernel (include/ttrnn.h: ttrnn_rnn_backward_ex) in the weight-gradient step.  A/B
# switch for tests and measurements; the library decides per descriptor what it can deliver.
USE_BWD_STATS = True
# Read the h_{t-1} rows of the hidden matrix's weight gradient in place (hints x_period / x_first) where the route can,
# instead of materialising [h_0, out[:, :-1]]; same kind of switch
USE_ROW_SHIFT = True
# Weight gradients through the chain kernel where the library offers it (ttrnn_rnn_wgrad, ABI 7); same kind of switch
USE_CHAIN_WGRAD = True
# tests: a list here receives (mask, stats, d_gates_in, d_gates_hid) of every layer backward
DEBUG_BWD_STATS = None

_ONES = {}


def _ones(dev, n):
    """fp32 ones[n] on `dev` (column bounds of rows that are hidden states: |h| <= 1), cached."""
    key = (dev, n)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(n, dtype=torch.float32, device=dev)
    return t


def _ttlinear_backward(spec, packed, x2d, dy2d, need_dx, need_dw, need_db, zeroed=None, hints=None):
    """zeroed: optional (dpk, db) fp32 buffers the caller has already zero-filled (one fill for several calls).
    hints: optional dict (struct ttrnn_lin_hints): fp32 device tensors x_colmax[in] / dy_colmax[out] / xdy_sum[out];
    x_period (int) + x_first (tensor or None): x2d holds a layer's outputs and the operand rows are the previous steps'."""
    lib = _lib.load()
    n = dy2d.shape[0]
    dev = dy2d.device
    dx = _alloc((n, spec.in_features), x2d.dtype, dev) if need_dx else None
    if zeroed is not None:
        dpk = zeroed[0] if need_dw else None
        db = zeroed[1] if need_db else None
    else:
        dpk = torch.zeros(spec.packed_elems, dtype=torch.float32, device=dev) if need_dw else None
        db = torch.zeros(spec.out_features, dtype=torch.float32, device=dev) if need_db else None
    wsb = lib.ttrnn_ttlinear_workspace(ctypes.byref(spec.desc), n)
    ws = _workspace(wsb, dev)
    hp = None
    if hints:
        for k, width in (("x_colmax", spec.in_features), ("dy_colmax", spec.out_features), ("xdy_sum", spec.out_features)):
            t = hints.get(k)
            if t is not None and (t.dtype != torch.float32 or t.numel() != width or not t.is_contiguous() or t.device != dev):
                raise ValueError("hint {} must be a contiguous fp32 tensor of {} elements on {}".format(k, width, dev))
        first = hints.get("x_first")
        if first is not None and (first.dtype != x2d.dtype or not first.is_contiguous() or first.device != dev):
            raise ValueError("hint x_first must be a contiguous {} tensor on {}".format(x2d.dtype, dev))
        rowmax = hints.get("dy_rowmax")
        if rowmax is not None and (rowmax.dtype != torch.float32 or rowmax.numel() != n or not rowmax.is_contiguous()
                                   or rowmax.device != dev):
            raise ValueError("hint dy_rowmax must be a contiguous fp32 tensor of {} elements on {}".format(n, dev))
        hs = _lib.LinHints(_ptr(hints.get("x_colmax")), _ptr(hints.get("dy_colmax")), _ptr(hints.get("xdy_sum")),
                           int(hints.get("x_period") or 0), _ptr(first), _ptr(rowmax))
        hp = ctypes.byref(hs)
    check(lib.ttrnn_ttlinear_backward_hinted(ctypes.byref(spec.desc), _dtype_code(x2d), _dtype_code(dy2d), n, _ptr(packed),
                                             _ptr(x2d), _ptr(dy2d), _ptr(dx), _ptr(dpk), _ptr(db), hp, _ptr(ws), wsb,
                                             _stream(dy2d)), "ttrnn_ttlinear_backward")
    return dx, dpk, db


class _TTLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, bias, spec, *cores):
        lib = _lib.load()
        packed = spec.pack(cores)
        n = x2d.shape[0]
        y = _alloc((n, spec.out_features), x2d.dtype, x2d.device)
        wsb = lib.ttrnn_ttlinear_workspace(ctypes.byref(spec.desc), n)
        ws = _workspace(wsb, x2d.device)
        check(lib.ttrnn_ttlinear_forward(ctypes.byref(spec.desc), _dtype_code(x2d), n, _ptr(packed), _ptr(bias),
                                         _ptr(x2d), _ptr(y), _ptr(ws), wsb, _stream(x2d)), "ttrnn_ttlinear_forward")
        ctx.spec = spec
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x2d, packed, *cores)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, packed = ctx.saved_tensors[:2]
        cores = ctx.saved_tensors[2:]
        spec = ctx.spec
        need_dx = ctx.needs_input_grad[0]
        need_db = ctx.has_bias and ctx.needs_input_grad[1]
        need_dw = any(ctx.needs_input_grad[3:])
        dy = dy.contiguous()
        dx, dpk, db = _ttlinear_backward(spec, packed, x2d, dy, need_dx, need_dw, need_db)
        dcores = spec.unpack_grads(dpk, cores) if need_dw else [None] * len(cores)
        if db is not None:
            db = db.to(dy.dtype)
        return (dx, db, None) + tuple(dcores)


class _TTHeadFn(torch.autograd.Function):
    """TTLinear + row-wise epilogue in one library call (ttrnn_head_forward / _backward)."""

    @staticmethod
    def forward(ctx, x2d, bias, spec, epi, *cores):
        lib = _lib.load()
        packed = spec.pack(cores)
        n = x2d.shape[0]
        y = _alloc((n, spec.out_features), x2d.dtype, x2d.device)
        aux = _alloc((n,), torch.float32, x2d.device)
        wsb = lib.ttrnn_head_workspace(ctypes.byref(spec.desc), n)
        ws = _workspace(wsb, x2d.device)
        check(lib.ttrnn_head_forward(ctypes.byref(spec.desc), _dtype_code(x2d), epi, n, _ptr(packed), _ptr(bias), _ptr(x2d),
                                     _ptr(y), _ptr(aux), _ptr(ws), wsb, _stream(x2d)), "ttrnn_head_forward")
        ctx.spec, ctx.epi, ctx.has_bias = spec, epi, bias is not None
        ctx.save_for_backward(x2d, packed, y, aux, *cores)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2d, packed, y, aux = ctx.saved_tensors[:4]
        cores = ctx.saved_tensors[4:]
        spec = ctx.spec
        need_dx = ctx.needs_input_grad[0]
        need_db = ctx.has_bias and ctx.needs_input_grad[1]
        need_dw = any(ctx.needs_input_grad[4:])
        n = x2d.shape[0]
        dev = x2d.device
        dy = dy.contiguous().to(x2d.dtype)
        dx = _alloc((n, spec.in_features), x2d.dtype, dev) if need_dx else None
        # the two accumulate-into buffers: ONE zero fill (each starts on a 256-byte boundary)
        npk = (spec.packed_elems + 63) // 64 * 64 if need_dw else 0
        flat = torch.zeros(npk + (spec.out_features if need_db else 0), dtype=torch.float32, device=dev)
        dpk = flat[:spec.packed_elems] if need_dw else None
        db = flat[npk:npk + spec.out_features] if need_db else None
        wsb = lib.ttrnn_head_workspace(ctypes.byref(spec.desc), n)
        ws = _workspace(wsb, dev)
        check(lib.ttrnn_head_backward(ctypes.byref(spec.desc), _dtype_code(x2d), ctx.epi, n, _ptr(packed), _ptr(x2d), _ptr(y),
                                      _ptr(aux), _ptr(dy), _ptr(dx), _ptr(dpk), _ptr(db), _ptr(ws), wsb, _stream(dy)),
              "ttrnn_head_backward")
        dcores = spec.unpack_grads(dpk, cores) if need_dw else [None] * len(cores)
        if db is not None:
            db = db.to(dy.dtype)
        return (dx, db, None, None) + tuple(dcores)


def tt_linear_head(x, cores, bias=None, spec=None, epilogue="log_softmax"):
    """TTLinear followed by the row-wise epilogue of one of the reference's two callers, fused into one library call:
    "log_softmax" (mnist_classifier.py:55-57) or "relu_l2norm" (speaker_encoder.py:86-89: ReLU, then L2 normalisation)."""
    if epilogue not in _lib.EPILOGUES or epilogue is None:
        raise ValueError("epilogue must be 'log_softmax' or 'relu_l2norm', got {!r}".format(epilogue))
    cores = list(cores)
    _require_device(x, bias, *cores)
    if spec is None:
        spec = TTSpec.from_cores(cores)
    if x.shape[-1] != spec.in_features:
        raise ValueError('Arguments shapes should align got {} and {} instead.'.format(
            [spec.out_features, spec.in_features], list(x.shape)))
    _check_operands(x, [("bias", bias)] + [("core {}".format(k), c) for k, c in enumerate(cores)])
    lead = x.shape[:-1]
    x2d = x.reshape(-1, spec.in_features).contiguous()
    with torch.cuda.device(x.device):
        y = _TTHeadFn.apply(x2d, bias, spec, _lib.EPILOGUES[epilogue], *cores)
    return y.reshape(*lead, spec.out_features)


def tt_linear(x, cores, bias=None, spec=None):
    """y[..., out] = TT(cores) x[..., in] + bias — the reference's TTLinear.forward."""
    cores = list(cores)
    _require_device(x, bias, *cores)
    if spec is None:
        spec = TTSpec.from_cores(cores)
    if x.shape[-1] != spec.in_features:
        raise ValueError('Arguments shapes should align got {} and {} instead.'.format(
            [spec.out_features, spec.in_features], list(x.shape)))
    _check_operands(x, [("bias", bias)] + [("core {}".format(k), c) for k, c in enumerate(cores)])
    lead = x.shape[:-1]
    x2d = x.reshape(-1, spec.in_features).contiguous()
    with torch.cuda.device(x.device):
        y = _TTLinearFn.apply(x2d, bias, spec, *cores)
    return y.reshape(*lead, spec.out_features)


class RnnLayerSpec(object):
    """Static description of one recurrent layer (cell kind + its two TT matrices)."""

    def __init__(self, cell, input_size, hidden_size, in_spec, hid_spec, has_bias_in, has_bias_hid, hid_blocks=1):
        assert cell in ("lstm", "gru")
        # hid_blocks = n_gates: the hidden matrix is TTLinearSet.joint_cores (block-diagonal cores behind a gate-selector core)
        self.hid_blocks = int(hid_blocks)
        self.cell = cell
        self.n_gates = 4 if cell == "lstm" else 3
        self.input_size = int(input_size)
        self.hidden_size = int(hidden_size)
        self.in_spec = in_spec
        self.hid_spec = hid_spec
        self.has_bias_in = bool(has_bias_in)
        self.has_bias_hid = bool(has_bias_hid)
        self.recording = True       # set per call by tt_rnn_layer: is autograd recording?
        gh = self.n_gates * self.hidden_size
        if in_spec.in_features != self.input_size or hid_spec.in_features != self.hidden_size or \
                in_spec.out_features != gh or hid_spec.out_features != gh:
            raise ValueError("TT shapes {}x{} / {}x{} do not match a {} layer in={} H={}".format(
                in_spec.in_modes, in_spec.out_modes, hid_spec.in_modes, hid_spec.out_modes, cell,
                input_size, hidden_size))

    def desc(self, batch, seq_len, dtype_code):
        d = RnnDesc()
        d.cell = TTRNN_LSTM if self.cell == "lstm" else TTRNN_GRU
        d.dtype = dtype_code
        d.batch, d.seq_len = int(batch), int(seq_len)
        d.input_size, d.hidden_size = self.input_size, self.hidden_size
        d.has_bias_in, d.has_bias_hid = int(self.has_bias_in), int(self.has_bias_hid)
        d.in_w = self.in_spec.desc
        d.hid_w = self.hid_spec.desc
        d.hid_blocks = self.hid_blocks
        return d


class _TTRnnLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, c0, bias_in, bias_hid, spec, n_in, stats, *cores):
        lib = _lib.load()
        cores_in, cores_hid = cores[:n_in], cores[n_in:]
        B, T, _ = x.shape
        H = spec.hidden_size
        dev = x.device
        desc = spec.desc(B, T, _dtype_code(x))
        # packing + forward as ONE C call (ABI 6: ttrnn_rnn_forward_cores — where the route has a fused set-up launch, two
        # launches per forward instead of five; elsewhere exactly ttrnn_pack_cores2 + ttrnn_rnn_forward)
        _require_device(*(list(cores_in) + list(cores_hid)))
        packed_in = _alloc((spec.in_spec.packed_elems,), torch.float32, dev)
        packed_hid = _alloc((spec.hid_spec.packed_elems,), torch.float32, dev)
        ptrs_in, strides_in = spec.in_spec._core_args(cores_in)
        ptrs_hid, strides_hid = spec.hid_spec._core_args(cores_hid)
        out = _alloc((B, T, H), x.dtype, dev)
        hT = _alloc((B, H), x.dtype, dev)
        cT = _alloc((B, H), x.dtype, dev) if spec.cell == "lstm" else None
        # needs_input_grad is True for Parameters even under torch.no_grad() (and grad mode is always
        # off inside forward): tt_rnn_layer decides outside whether a graph is being recorded, so that
        # inference does not pay for the training reserve (cfg2: 411 MB of writes per forward)
        need_grad = spec.recording and any(ctx.needs_input_grad)
        reserve = None
        if need_grad or (stats is not None and spec.cell == "lstm"):      # the per-step c_t live in the reserve records
            reserve = _alloc((lib.ttrnn_rnn_reserve_bytes(ctypes.byref(desc)) // 4,), torch.float32, dev)
        wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
        ws = _workspace(wsb, dev)
        with _timed("ttrnn_rnn_forward"):
            check(lib.ttrnn_rnn_forward_cores(ctypes.byref(desc), _ptr(x), _ptr(h0), _ptr(c0), ptrs_in, strides_in,
                                              _ptr(bias_in), ptrs_hid, strides_hid, _ptr(bias_hid), _ptr(packed_in),
                                              _ptr(packed_hid), _ptr(out), _ptr(hT), _ptr(cT), _ptr(reserve), _ptr(ws), wsb,
                                              _stream(x)), "ttrnn_rnn_forward_cores")
        if stats is not None:
            # LSTM reserve: gates [B][T][H][4], then the cell states [B][T][H] (ttrnn_core.h: res_gate / res_cell)
            stats.forward(out, reserve[4 * B * T * H:].view(B, T, H) if spec.cell == "lstm" else None)
        ctx.stats = stats
        ctx.spec = spec
        # an output nobody differentiates (the classifier consumes hT only) arrives as None in backward instead of a zero tensor
        # torch would allocate and fill (cfg5: 537 MB, 78 us per step; the library takes NULL for "no gradient")
        ctx.set_materialize_grads(False)
        ctx.x_bounded = bool(getattr(spec, "x_bounded", False))
        ctx.n_in = n_in
        ctx.flags = (h0 is not None, c0 is not None, bias_in is not None, bias_hid is not None)
        saved = [x, out, reserve, packed_in, packed_hid]
        saved += [t for t in (h0, c0) if t is not None]
        ctx.save_for_backward(*(saved + list(cores)))
        if spec.cell == "lstm":
            return out, hT, cT
        return out, hT

    @staticmethod
    def backward(ctx, d_out, d_hT, d_cT=None):
        lib = _lib.load()
        spec = ctx.spec
        has_h0, has_c0, has_bin, has_bhid = ctx.flags
        saved = list(ctx.saved_tensors)
        x, out, reserve, packed_in, packed_hid = saved[:5]
        pos = 5
        h0 = c0 = None
        if has_h0:
            h0 = saved[pos]; pos += 1
        if has_c0:
            c0 = saved[pos]; pos += 1
        cores = saved[pos:]
        cores_in, cores_hid = cores[:ctx.n_in], cores[ctx.n_in:]
        B, T, _ = x.shape
        H, G = spec.hidden_size, spec.n_gates
        dev = x.device
        desc = spec.desc(B, T, _dtype_code(x))
        d_out = d_out.contiguous() if d_out is not None else None
        d_hT = d_hT.contiguous() if d_hT is not None else None
        d_cT = d_cT.contiguous() if d_cT is not None else None
        dg_in = _alloc((B, T, G * H), torch.float32, dev)
        dg_hid = _alloc((B, T, G * H), torch.float32, dev) if spec.cell == "gru" else dg_in
        need = ctx.needs_input_grad
        d_h0 = _alloc((B, H), x.dtype, dev) if (has_h0 and need[1]) else None
        d_c0 = _alloc((B, H), x.dtype, dev) if (has_c0 and need[2]) else None
        wsb = lib.ttrnn_rnn_backward_workspace_ex(ctypes.byref(desc), int(ctx.stats is not None))
        ws = _workspace(wsb, dev)
        d_state = _alloc((B, T, H, 2), torch.float32, dev) if ctx.stats is not None else None
        # by-products of the reverse-time kernel for the weight-gradient step below (column maxima of the gate gradients;
        # input_size == 1: the sums over the rows that ARE that layer's input-weight and bias gradients)
        mask = lib.ttrnn_rnn_backward_stats(ctypes.byref(desc)) if (USE_BWD_STATS and d_state is None) else 0
        # (TTRNN_BWD_STATS_ROWMAX: the rows' maxima of d_gates_in behind the four rows — the dx GEMM's row scales)
        rowmax = None
        if mask & _lib.BWD_STATS_ROWMAX:
            sflat = _alloc((_lib.BWD_STATS_ROWS * G * H + B * T,), torch.float32, dev)
            bstats = sflat[:_lib.BWD_STATS_ROWS * G * H].view(_lib.BWD_STATS_ROWS, G * H)
            rowmax = sflat[_lib.BWD_STATS_ROWS * G * H:]
        else:
            bstats = _alloc((_lib.BWD_STATS_ROWS, G * H), torch.float32, dev) if mask else None
        in1 = bool(mask & _lib.BWD_STATS_IN1SUMS)
        with _timed("ttrnn_rnn_backward"):
            check(lib.ttrnn_rnn_backward_ex(ctypes.byref(desc), _ptr(out), _ptr(h0), _ptr(c0), _ptr(packed_hid),
                                            _ptr(reserve), _ptr(d_out), _ptr(d_hT), _ptr(d_cT), _ptr(dg_in),
                                            _ptr(dg_hid), _ptr(d_h0), _ptr(d_c0), _ptr(d_state),
                                            _ptr(x) if in1 else ctypes.c_void_p(0), _ptr(bstats), _ptr(ws), wsb,
                                            _stream(x)), "ttrnn_rnn_backward")
        if d_state is not None:
            ctx.stats.backward(d_state[..., 0], d_state[..., 1] if spec.cell == "lstm" else None)
        if DEBUG_BWD_STATS is not None:
            DEBUG_BWD_STATS.append((mask, bstats, dg_in, dg_hid, rowmax))
        # weight / input gradients: two TTLinear backward passes over the B*T rows
        n_in = ctx.n_in
        need_dw_in = any(need[8:8 + n_in])
        need_dw_hid = any(need[8 + n_in:])
        # the accumulate-into buffers of both matrices (packed core gradients, bias gradients): ONE zero fill
        sizes = [spec.in_spec.packed_elems, G * H, spec.hid_spec.packed_elems, G * H]
        padded = [(n + 63) // 64 * 64 for n in sizes]                       # every buffer starts on a 256-byte boundary
        flat = torch.zeros(sum(padded), dtype=torch.float32, device=dev)
        z_in_w, z_in_b, z_hid_w, z_hid_b = (t[:n] for t, n in zip(torch.split(flat, padded), sizes))
        hints_in = hints_hid = None
        want_db_in = has_bin and need[3]
        if mask & _lib.BWD_STATS_COLMAX:
            # rows that are hidden states: |h_t| <= 1 for t >= 1 (LSTM: o * tanh(c); GRU: convex combinations of tanh values
            # and h_{t-1}, so max(1, |h_0|) per unit) — and the caller's h_0 is row 0 of every sample
            state_bound = _ones(dev, H) if h0 is None else torch.maximum(h0.float().abs().amax(0), _ones(dev, H))
            hints_in = {"dy_colmax": bstats[0], "x_colmax": _ones(dev, spec.input_size) if ctx.x_bounded else None}
            hints_hid = {"dy_colmax": bstats[1], "x_colmax": state_bound}
            if in1 and not need[0]:
                hints_in["xdy_sum"] = bstats[2]
            if rowmax is not None and need[0]:
                hints_in["dy_rowmax"] = rowmax
        use_sums = bool(hints_in and hints_in.get("xdy_sum") is not None)
        want_db_hid = has_bhid and need[4]
        # ABI 7: both weight gradients through the CHAIN where it is the cheaper contraction (low ranks on large modes — the
        # reference's speaker encoder, params_model.py:2-4,14-16): one pass over the gate gradients for both matrices of an
        # LSTM layer.  `chain` = the matrices it took (bit 0 input, bit 1 hidden); everything else keeps the dense routes below.
        chain = 0
        if USE_CHAIN_WGRAD and need_dw_hid and T > 0 and x.dtype == torch.float32 and x.is_contiguous():
            cands = ((3, 2) if (need_dw_in and not need[0] and spec.cell == "lstm") else (2,))
            for cand in cands:
                cwsb = lib.ttrnn_rnn_wgrad_workspace(ctypes.byref(desc), cand)
                if cwsb > 0:
                    chain = cand
                    break
        if chain:
            cws = _workspace(cwsb, dev)
            hb = _ones(dev, H) if h0 is None else torch.maximum(h0.float().abs().amax(0), _ones(dev, H))
            cm = bool(mask & _lib.BWD_STATS_COLMAX)
            wa = _lib.WgradArgs(
                _ptr(x), _ptr(out), _ptr(h0), _ptr(dg_in), _ptr(dg_hid), _ptr(packed_in), _ptr(packed_hid),
                _ptr(z_in_w) if chain & 1 else None, _ptr(z_hid_w),
                _ptr(z_in_b) if (chain & 1 and want_db_in) else None, _ptr(z_hid_b) if want_db_hid else None,
                _ptr(_ones(dev, spec.input_size)) if ctx.x_bounded else None, _ptr(hb),
                _ptr(bstats[0]) if cm else None, _ptr(bstats[1]) if cm else None)
            with _timed("ttrnn_rnn_wgrad"):
                check(lib.ttrnn_rnn_wgrad(ctypes.byref(desc), chain, ctypes.byref(wa), _ptr(cws), cwsb, _stream(x)),
                      "ttrnn_rnn_wgrad")
        dx = dpk_in = db_in = None
        if chain & 1:
            dpk_in, db_in = z_in_w, (z_in_b if want_db_in else None)
        else:
            dx, dpk_in, db_in = _ttlinear_backward(spec.in_spec, packed_in, x.reshape(B * T, -1),
                                                   dg_in.reshape(B * T, -1), need[0], need_dw_in,
                                                   want_db_in and not use_sums, zeroed=(z_in_w, z_in_b), hints=hints_in)
            if use_sums and want_db_in:
                db_in = bstats[3]
        # h_{t-1} rows: [h0, out[:, :-1]] — read in place by the dense-gradient routes, materialised for the others
        if chain & 2:
            dpk_hid, db_hid = z_hid_w, (z_hid_b if want_db_hid else None)
        elif USE_ROW_SHIFT and need_dw_hid and T > 0 and lib.ttrnn_ttlinear_backward_shift_ok(
                ctypes.byref(spec.hid_spec.desc), _dtype_code(out), _lib.TTRNN_F32, B * T, T, 0):
            hprev = out.reshape(B * T, H)
            hints_hid = dict(hints_hid or {}, x_period=T, x_first=h0)
        else:
            first = h0 if h0 is not None else torch.zeros(B, H, dtype=out.dtype, device=dev)
            hprev = torch.cat([first.unsqueeze(1), out[:, :-1]], dim=1).reshape(B * T, H)
        if not chain & 2:
            _, dpk_hid, db_hid = _ttlinear_backward(spec.hid_spec, packed_hid, hprev, dg_hid.reshape(B * T, -1),
                                                    False, need_dw_hid, want_db_hid, zeroed=(z_hid_w, z_hid_b),
                                                    hints=hints_hid)
        dcin = spec.in_spec.unpack_grads(dpk_in, cores_in) if need_dw_in else [None] * n_in
        dchid = spec.hid_spec.unpack_grads(dpk_hid, cores_hid) if need_dw_hid else [None] * len(cores_hid)
        if dx is not None:
            dx = dx.reshape(x.shape)
        if db_in is not None:
            db_in = db_in.to(x.dtype)
        if db_hid is not None:
            db_hid = db_hid.to(x.dtype)
        return (dx, d_h0, d_c0, db_in, db_hid, None, None, None) + tuple(dcin) + tuple(dchid)


class PreparedLayer(object):
    """Weight-only state of one recurrent layer kept across no-grad forwards (include/ttrnn.h: ttrnn_rnn_forward_phase):
    the packed cores, and per (batch, seq_len, dtype) a workspace a TTRNN_PHASE_PREPARE call has filled with everything
    that depends on the weights alone — the fused core's scale header and fragments and, for input_size == 1, the input
    projection of the unit rows.  A forward is then ONE launch (the recurrent kernel) instead of five.

    Opt-in (FusedRnnBase.prepare_for_inference): a stale copy would be a silent wrong answer.  In-place updates of the
    parameters (optimizer steps, load_state_dict, .copy_) bump their version counters and are detected — the layer is
    prepared again; writes through `.data` are NOT tracked by torch: call prepare_for_inference() again after them."""

    MAX_WORKSPACES = 4      # distinct (batch, seq_len, dtype) shapes kept per layer (least recently used goes first)

    def __init__(self, spec, cores_in, bias_in, cores_hid, bias_hid, params=None):
        """params: the nn.Parameters the operands derive from — what the freshness stamp watches.  For TTLinear / nn.Linear
        weights the operands ARE (views of) the parameters; a TTLinearSet's joint cores are fresh torch.cat copies whose
        version counters never move, so the set's own per-gate parameters must be given here."""
        self.spec = spec
        operands = list(cores_in) + list(cores_hid) + [b for b in (bias_in, bias_hid) if b is not None]
        self.tensors = list(params) if params is not None else operands
        self.stamp = self._stamp()
        with torch.no_grad(), torch.cuda.device(operands[0].device):
            self.packed_in, self.packed_hid = TTSpec.pack_pair(spec.in_spec, list(cores_in), spec.hid_spec, list(cores_hid))
        self.workspaces = collections.OrderedDict()

    def _stamp(self):
        return tuple((t.data_ptr(), t._version, t.dtype, t.device) for t in self.tensors)

    def fresh(self):
        return self.stamp == self._stamp()


def _rnn_forward_prepared(spec, x, h0, c0, bias_in, bias_hid, prep, need_out=True):
    """No-grad forward of one layer on prepared weights: TTRNN_PHASE_PREPARE once per (B, T, dtype, options), then
    TTRNN_PHASE_RUN per call."""
    lib = _lib.load()
    B, T, _ = x.shape
    H = spec.hidden_size
    dev = x.device
    dt = _dtype_code(x)
    desc = spec.desc(B, T, dt)
    if lib.ttrnn_rnn_prepare_supported(ctypes.byref(desc)):
        # the workspace holds weight-only results a PREPARE call left there: kept per shape, a few shapes at most
        key = (B, T, dt, dev, _lib.OPTIONS_EPOCH)
        ent = prep.workspaces.get(key)
        if ent is None:
            wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
            ws = _workspace(wsb, dev)
            null = ctypes.c_void_p(0)
            check(lib.ttrnn_rnn_forward_phase(ctypes.byref(desc), _lib.PHASE_PREPARE, null, null, null, _ptr(prep.packed_in),
                                              _ptr(bias_in), _ptr(prep.packed_hid), _ptr(bias_hid), null, null, null, null,
                                              _ptr(ws), wsb, _stream(x)), "ttrnn_rnn_forward_phase(PREPARE)")
            for k in [k for k in prep.workspaces if k[4] != _lib.OPTIONS_EPOCH]:                     # drop stale epochs
                del prep.workspaces[k]
            while len(prep.workspaces) >= prep.MAX_WORKSPACES:
                prep.workspaces.popitem(last=False)
            ent = prep.workspaces[key] = (ws, wsb)
        else:
            prep.workspaces.move_to_end(key)
        ws, wsb = ent
    else:
        # this route has no weight-only part in its workspace (PREPARE is a no-op, RUN does everything): the buffer is per-call
        # scratch — e.g. the [B][T][H][4] projections of a stacked layer, hundreds of MB — and is NOT kept
        wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
        ws = _workspace(wsb, dev)
    want_out = need_out or not lib.ttrnn_rnn_out_optional(ctypes.byref(desc))
    out = _alloc((B, T, H), x.dtype, dev) if want_out else None
    hT = _alloc((B, H), x.dtype, dev)
    cT = _alloc((B, H), x.dtype, dev) if spec.cell == "lstm" else None
    with _timed("ttrnn_rnn_forward"):
        check(lib.ttrnn_rnn_forward_phase(ctypes.byref(desc), _lib.PHASE_RUN, _ptr(x), _ptr(h0), _ptr(c0),
                                          _ptr(prep.packed_in), _ptr(bias_in), _ptr(prep.packed_hid), _ptr(bias_hid),
                                          _ptr(out), _ptr(hT), _ptr(cT), ctypes.c_void_p(0), _ptr(ws), wsb, _stream(x)),
              "ttrnn_rnn_forward_phase(RUN)")
    return (out, hT, cT) if spec.cell == "lstm" else (out, hT)


def rnn_route(spec, batch, seq_len, dtype=torch.float32):
    """Name of the kernel family ttrnn_rnn_forward runs for this layer at this size ("fused_core", "runtime_mfma", "valu", ...)."""
    desc = spec.desc(batch, seq_len, _DT[dtype])
    code = _lib.load().ttrnn_rnn_forward_route(ctypes.byref(desc))
    if code < 0:
        check(code, "ttrnn_rnn_forward_route")
    return _lib.ROUTES[code]


def rnn_samples_per_workgroup(spec, batch, seq_len, dtype=torch.float32):
    """Samples a workgroup of the recurrent forward kernel carries for this layer at this size (2: the tier's paired kernel)."""
    desc = spec.desc(batch, seq_len, _DT[dtype])
    n = _lib.load().ttrnn_rnn_forward_samples_per_workgroup(ctypes.byref(desc))
    if n < 0:
        check(n, "ttrnn_rnn_forward_samples_per_workgroup")
    return n


def rnn_backward_route(spec, batch, seq_len, dtype=torch.float32, want_state=False):
    """Name of the kernel family of the reverse-time kernel (BPTT) for this layer at this size."""
    desc = spec.desc(batch, seq_len, _DT[dtype])
    code = _lib.load().ttrnn_rnn_backward_route(ctypes.byref(desc), 1 if want_state else 0)
    if code < 0:
        check(code, "ttrnn_rnn_backward_route")
    return _lib.ROUTES[code]


class StepStats(object):
    """Feeds the per-timestep statistics of ActivGradLogger (reference: tensorized_rnn/rnn_utils.py:127-171,217-226 —
    mean over the batch of ||v_t||^2 and of log ||v_t||^2, for v = h, c and their gradients) from ONE fused sequence call
    instead of T per-step cell calls: activations from `out` and the reserve's c_t, gradients from the d_state output of
    the reverse-time kernel.  h_logger / c_logger are ActivGradLogger instances (c_logger None for GRU)."""

    def __init__(self, h_logger, c_logger=None):
        self.h_logger, self.c_logger = h_logger, c_logger

    @staticmethod
    def _av(v):                          # [B, T, H] -> (mean_b ||.||^2 [T], mean_b log ||.||^2 [T])
        n = v.float().square().sum(2)
        return n.mean(0), n.log().mean(0)

    @torch.no_grad()
    def forward(self, out, c_steps):
        for lg, v in ((self.h_logger, out), (self.c_logger, c_steps)):
            if lg is None or v is None:
                continue
            av, avl = self._av(v)
            lg.act.extend(av.unbind(0))
            lg.log_act.extend(avl.unbind(0))

    @torch.no_grad()
    def backward(self, dh, dc):
        for lg, v in ((self.h_logger, dh), (self.c_logger, dc)):
            if lg is None or v is None:
                continue
            av, avl = self._av(v)
            for a, b in zip(reversed(av.unbind(0)), reversed(avl.unbind(0))):      # the hooks fire last timestep first
                lg.grad.appendleft(a)
                lg.log_grad.appendleft(b)


def _rnn_forward_nograd(spec, x, h0, c0, cores_in, bias_in, cores_hid, bias_hid, need_out):
    """No-grad forward without the autograd Function (nothing is saved): lets the last layer skip `out` where the route can."""
    lib = _lib.load()
    B, T, _ = x.shape
    H = spec.hidden_size
    dev = x.device
    desc = spec.desc(B, T, _dtype_code(x))
    packed_in, packed_hid = TTSpec.pack_pair(spec.in_spec, cores_in, spec.hid_spec, cores_hid)
    want_out = need_out or not lib.ttrnn_rnn_out_optional(ctypes.byref(desc))
    out = _alloc((B, T, H), x.dtype, dev) if want_out else None
    hT = _alloc((B, H), x.dtype, dev)
    cT = _alloc((B, H), x.dtype, dev) if spec.cell == "lstm" else None
    wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
    ws = _workspace(wsb, dev)
    with _timed("ttrnn_rnn_forward"):
        check(lib.ttrnn_rnn_forward(ctypes.byref(desc), _ptr(x), _ptr(h0), _ptr(c0), _ptr(packed_in), _ptr(bias_in),
                                    _ptr(packed_hid), _ptr(bias_hid), _ptr(out), _ptr(hT), _ptr(cT), ctypes.c_void_p(0),
                                    _ptr(ws), wsb, _stream(x)), "ttrnn_rnn_forward")
    return (out, hT, cT) if spec.cell == "lstm" else (out, hT)


def tt_rnn_layer(spec, x, h0, c0, cores_in, bias_in, cores_hid, bias_hid, stats=None, prepared=None, need_out=True,
                 x_bounded=False):
    """One recurrent layer over the whole sequence on the device.
    Returns (out[B,T,H], hT[B,H], cT[B,H]) for LSTM and (out, hT) for GRU.
    stats: optional StepStats — ActivGradLogger's per-step statistics without leaving the fused path.
    prepared: optional PreparedLayer — used when autograd is not recording (inference on unchanged weights).
    need_out=False (no-grad calls only): the caller consumes only the final state; `out` is returned as None where the
    route can skip it (include/ttrnn.h: ttrnn_rnn_out_optional), computed as usual elsewhere.
    x_bounded=True: the caller guarantees |x| <= 1 everywhere (x is the output of a recurrent layer below): the input matrix's
    weight gradient then needs no pass over x for its column scales (ttrnn_ttlinear_backward_hinted)."""
    cores_in, cores_hid = list(cores_in), list(cores_hid)
    _require_device(x, h0, c0, bias_in, bias_hid, *(cores_in + cores_hid))
    if x.dim() != 3 or x.shape[2] != spec.input_size:
        raise ValueError("expected input of shape (batch, seq_len, {}), got {}".format(spec.input_size, tuple(x.shape)))
    _check_operands(x, [("bias_in", bias_in), ("bias_hid", bias_hid)] + [("core", c) for c in cores_in + cores_hid])
    for name, t in (("h0", h0), ("c0", c0)):
        if t is not None and t.device != x.device:
            raise _lib.TtrnnError("{} is on {} but the input is on {}".format(name, t.device, x.device))
    x = x.contiguous()
    h0 = h0.contiguous().to(x.dtype) if h0 is not None else None
    c0 = c0.contiguous().to(x.dtype) if (c0 is not None and spec.cell == "lstm") else None
    spec.recording = torch.is_grad_enabled()
    spec.x_bounded = bool(x_bounded)
    if prepared is not None and not spec.recording and stats is None and POISON_ALLOCATIONS is False:
        with torch.cuda.device(x.device):
            return _rnn_forward_prepared(spec, x, h0, c0, bias_in, bias_hid, prepared, need_out)
    if not need_out and not spec.recording and stats is None:
        with torch.cuda.device(x.device):
            return _rnn_forward_nograd(spec, x, h0, c0, cores_in, bias_in, cores_hid, bias_hid, need_out)
    with torch.cuda.device(x.device):       # the library launches on the CURRENT device's context
        return _TTRnnLayerFn.apply(x, h0, c0, bias_in, bias_hid, spec, len(cores_in), stats, *(cores_in + cores_hid))
