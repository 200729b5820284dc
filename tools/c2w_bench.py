"""Times ttrnn_rnn_wgrad (the chain weight-gradient kernel, ttrnn_fast_c2w.hip) alone at the speaker-encoder size — B = 512,
T = 160, in = 40, H = 768, d = 2 (params_model.py) — on random operands:  python tools/c2w_bench.py [rank] [mats] [reps]
With TTRNN_LIB_PATH=tools/bin/libttrnn_abl.so and TTRNN_DEV2=<65536 * bits> the ablation build leaves out phases (bits: 1 A/B,
2 C/D, 4 staging stores, 8 global loads)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tensorized-rnn_amd")]
from ttrnn_hip import _lib, functional as F      # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B, T, H, n_in = 512, 160, 768, 40
lib = _lib.load()
d = torch.device("cuda:0")
inp = ([5, 8], [48, 64], [1, r, 1])
hid = ([24, 32], [48, 64], [1, r, 1])
spec_in, spec_hid = F.TTSpec(*inp), F.TTSpec(*hid)
spec = F.RnnLayerSpec("lstm", n_in, H, spec_in, spec_hid, True, True)
desc = spec.desc(B, T, _lib.TTRNN_F32)
wsb = lib.ttrnn_rnn_wgrad_workspace(ctypes.byref(desc), mats)
assert wsb > 0
g = torch.Generator(device=d).manual_seed(1)
mk = lambda modes: [torch.randn(modes[2][k], modes[0][k], modes[1][k], modes[2][k + 1], device=d, generator=g).transpose(1, 2) * 0.3
                    for k in range(len(modes[0]))]
ci, ch = mk(inp), mk(hid)
pk_in, pk_hid = spec_in.pack(ci), spec_hid.pack(ch)
x = torch.rand(B, T, n_in, device=d, generator=g)
out = torch.tanh(torch.randn(B, T, H, device=d, generator=g))
dg = torch.randn(B, T, 4 * H, device=d, generator=g)
ws = F._workspace(wsb, d)
dpi = torch.zeros(spec_in.packed_elems, device=d)
dph = torch.zeros(spec_hid.packed_elems, device=d)
dbi = torch.zeros(4 * H, device=d)
dbh = torch.zeros(4 * H, device=d)
hb = torch.ones(H, device=d)
xb = torch.ones(n_in, device=d)
dyb = dg.abs().amax((0, 1)).contiguous()
wa = _lib.WgradArgs(F._ptr(x), F._ptr(out), None, F._ptr(dg), F._ptr(dg), F._ptr(pk_in), F._ptr(pk_hid),
                    F._ptr(dpi) if mats & 1 else None, F._ptr(dph), F._ptr(dbi) if mats & 1 else None, F._ptr(dbh),
                    F._ptr(xb), F._ptr(hb), F._ptr(dyb), F._ptr(dyb))


def call():
    _lib.check(lib.ttrnn_rnn_wgrad(ctypes.byref(desc), mats, ctypes.byref(wa), F._ptr(ws), wsb, F._stream(x)), "ttrnn_rnn_wgrad")


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
print("rank %d mats %d dev2 %s: %.1f us per call (prep + chain kernel + reduce + bias)" % (
    r, mats, os.environ.get("TTRNN_DEV2", "0"), e0.elapsed_time(e1) * 1e3 / reps))

if "abl" in os.environ.get("TTRNN_LIB_PATH", ""):
    st = ws[64:64 + 128].view(torch.int64).cpu().tolist()
    nblk = ((B * T + 1) // 2 + 255) // 256
    names = ["stage stores", "prefetch + barrier", "phase A", "phase B", "barrier AB", "phases C D", "barrier CD", "loop top"]
    if not int(os.environ.get("TTRNN_DEV2", "0")) & 64:      # the register hand-off kernel (ttrnn_c2r_dev.h)
        names = ["load issue", "step 0", "step 1", "step 2", "step 3", "stage stores", "barrier", "loop top"]
    for w, off in ((0, 0), (5, 8)):
        print("wave %d cycles per block:" % w, ", ".join("%s %d" % (n, st[off + i] // max(nblk, 1)) for i, n in enumerate(names)),
              "| total", sum(st[off:off + 8]) // max(nblk, 1))
