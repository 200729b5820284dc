#!/usr/bin/env python3
"""Developer check: the fused set-up launch (ttrnn_fast_setup.hip) against the four launches it replaces — packed cores, unit rows,
scale header, fragments and the results, byte for byte, through the C ABI."""
import contextlib, ctypes, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import torch
import ttrnn_hip
from ttrnn_hip import _lib, functional as F
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU

dev = torch.device("cuda:0")
lib = _lib.load()
for cls in (TTLSTM, TTGRU):
    torch.manual_seed(int(os.environ.get('SEED', '5')))
    with contextlib.redirect_stdout(io.StringIO()):
        m = cls(1, 256, 1, dev, n_cores=3, tt_rank=8)
    layer = m._all_layers[0]
    spec = layer._layer_spec()
    B, T = 6, int(os.environ.get('TT', '33'))
    x = torch.randn(B, T, 1, device=dev)
    desc = spec.desc(B, T, 0)
    cores_in = list(layer.input_weights.parameters()) if hasattr(layer.input_weights, "parameters") else None
    ci = [p for n, p in layer.input_weights.named_parameters() if "parameters" in n]
    ch = [p for n, p in layer.hidden_weights.named_parameters() if "parameters" in n]
    bi, bh = layer.input_weights.bias, layer.hidden_weights.bias
    wsb = lib.ttrnn_rnn_workspace(ctypes.byref(desc))
    res = {}
    for name, d in (("fused", 0), ("separate", 65536)):
        with ttrnn_hip.option("dev", d):
            ws = (torch.randn(wsb // 4 + 16, device=dev) if os.environ.get('GARBAGE') else torch.zeros(wsb // 4 + 16, dtype=torch.float32, device=dev))
            pin = torch.zeros(spec.in_spec.packed_elems, device=dev)
            phid = torch.zeros(spec.hid_spec.packed_elems, device=dev)
            out = torch.zeros(B, T, 256, device=dev)
            hT = torch.zeros(B, 256, device=dev)
            cT = torch.zeros(B, 256, device=dev)
            pi, si = spec.in_spec._core_args(ci)
            ph, sh = spec.hid_spec._core_args(ch)
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            st = lib.ttrnn_rnn_forward_cores(ctypes.byref(desc), P(x), None, None, pi, si, P(bi), ph, sh, P(bh), P(pin), P(phid), P(out),
                                             P(hT), P(cT) if cls is TTLSTM else None, None, P(ws), wsb, None)
            torch.cuda.synchronize()
            assert st == 0, st
            res[name] = dict(ws=ws.cpu(), pin=pin.cpu(), phid=phid.cpu(), out=out.cpu(), hT=hT.cpu())
    a, b = res["fused"], res["separate"]
    print("  hT equal", torch.equal(a["hT"], b["hT"]))
    print(cls.__name__, "packed_in equal", torch.equal(a["pin"], b["pin"]), "packed_hid equal", torch.equal(a["phid"], b["phid"]),
          "out equal", torch.equal(a["out"], b["out"]))
    wa, wb = a["ws"].view(torch.int32), b["ws"].view(torch.int32)
    diff = (wa != wb).nonzero().flatten()
    print("  workspace words differing:", diff.numel(), "of", wa.numel(), "first:", diff[:12].tolist())
    gin_words = 2 * 256 * 4
    print("  gin rows differing:", int((wa[:gin_words] != wb[:gin_words]).sum()), " max abs", float((a["ws"][:gin_words] - b["ws"][:gin_words]).abs().max()))
