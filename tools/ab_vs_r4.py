#!/usr/bin/env python3
"""A/B of the headline forward against the round-4 library ("bit-identical outputs to today's"):
    tools/bin/libttrnn_r4.so  = `make` of commit 96adec3's tensorized-rnn_amd/csrc (build it there; tools/bin/ is not tracked)
Both libraries are driven through the C ABI subset they share (ttrnn_pack_cores2 + ttrnn_rnn_forward) on the same strided
Parameters and inputs; the current library additionally through ttrnn_rnn_forward_cores (the fused set-up launch)."""
import contextlib, ctypes, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import torch
from ttrnn_hip import _lib, functional as F
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU

dev = torch.device("cuda:0")
new = _lib.load()
old = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libttrnn_r4.so"))
for name in ("ttrnn_pack_cores2", "ttrnn_rnn_forward", "ttrnn_rnn_workspace"):
    fn = getattr(old, name)
    fn.restype, fn.argtypes = _lib._SIGNATURES[name]
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for label, cls, B, T, seeds in (("cfg2 TT-LSTM", TTLSTM, 64, 784, (1111, 8)), ("cfg1-size batch TT-LSTM", TTLSTM, 6, 30, (8,))):
    for seed in seeds:
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            m = cls(1, 256, 1, dev, n_cores=3, tt_rank=8)
        layer = m._all_layers[0]
        spec = layer._layer_spec()
        x = torch.rand(B, T, 1, device=dev)
        desc = spec.desc(B, T, 0)
        ci = [p for n, p in layer.input_weights.named_parameters() if "parameters" in n]
        ch = [p for n, p in layer.hidden_weights.named_parameters() if "parameters" in n]
        bi, bh = layer.input_weights.bias, layer.hidden_weights.bias
        pi, si = spec.in_spec._core_args(ci)
        ph, sh = spec.hid_spec._core_args(ch)
        res = {}
        for which, lib in (("r4", old), ("r5", new), ("r5_fused", new)):
            wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
            ws = torch.randn(wsb // 4 + 16, device=dev)
            pin = torch.randn(spec.in_spec.packed_elems, device=dev)
            phid = torch.randn(spec.hid_spec.packed_elems, device=dev)
            out = torch.empty(B, T, 256, device=dev)
            hT = torch.empty(B, 256, device=dev)
            cT = torch.empty(B, 256, device=dev)
            if which == "r5_fused":
                assert lib.ttrnn_rnn_forward_cores_fused(ctypes.byref(desc)) == 1
                st = lib.ttrnn_rnn_forward_cores(ctypes.byref(desc), P(x), None, None, pi, si, P(bi), ph, sh, P(bh), P(pin), P(phid), P(out),
                                                 P(hT), P(cT), None, P(ws), wsb, None)
            else:
                st = lib.ttrnn_pack_cores2(ctypes.byref(spec.in_spec.desc), pi, si, P(pin), ctypes.byref(spec.hid_spec.desc), ph, sh, P(phid), 0, None)
                assert st == 0, st
                st = lib.ttrnn_rnn_forward(ctypes.byref(desc), P(x), None, None, P(pin), P(bi), P(phid), P(bh), P(out), P(hT), P(cT), None,
                                           P(ws), wsb, None)
            torch.cuda.synchronize()
            assert st == 0, (which, st)
            res[which] = (out.clone(), hT.clone(), cT.clone())
        for which in ("r5", "r5_fused"):
            eq = all(torch.equal(a, b) for a, b in zip(res["r4"], res[which]))
            print("%s  B=%d T=%d seed %d: %s == r4 bit for bit: %s" % (label, B, T, seed, which, eq))
