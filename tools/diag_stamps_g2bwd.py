#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the runtime-shape reverse-time kernel k_g2_bwd (s_memtime stamps that exist only in
the -DTTRNN_ABLATIONS build: `make -C tensorized-rnn_amd/csrc ablation`, loaded through TTRNN_LIB_PATH).  Shares only — never quote
this build's run time.   python tools/diag_stamps_g2bwd.py [benchmarking.py-style flags: --gru --naive_tt --hidden_size H ...]"""
import argparse, contextlib, ctypes, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TTRNN_LIB_PATH", os.path.join(ROOT, "tools", "bin", "libttrnn_abl.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import numpy as np, torch
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU
from ttrnn_hip import functional as F, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--gru", action="store_true"); ap.add_argument("--naive_tt", action="store_true")
ap.add_argument("--in_size", type=int, default=256); ap.add_argument("--hidden_size", type=int, default=512)
ap.add_argument("--ncores", type=int, default=3); ap.add_argument("--ttrank", type=int, default=8)
ap.add_argument("--batch_size", type=int, default=512); ap.add_argument("--seq_len", type=int, default=160)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    cls = TTGRU if a.gru else TTLSTM
    m = cls(a.in_size, a.hidden_size, 1, dev, n_cores=a.ncores, tt_rank=a.ttrank, is_naive=a.naive_tt)
x = torch.rand(a.batch_size, a.seq_len, a.in_size, device=dev)
seen = []
orig_ws = F._workspace
def ws(nbytes, device):
    t = orig_ws(nbytes, device)
    seen.append((nbytes, t))
    return t
F._workspace = ws
spec = m._all_layers[0]._layer_spec()
print("backward route:", F.rnn_backward_route(spec, a.batch_size, a.seq_len))
desc = spec.desc(a.batch_size, a.seq_len, _lib.TTRNN_F32)
bws = _lib.load().ttrnn_rnn_backward_workspace(ctypes.byref(desc))
for _ in range(2):
    seen.clear()
    out = m(x)[0]
    out.sum().backward()
torch.cuda.synchronize()
buf = next(t for n, t in seen if n == bws)
raw = buf.view(torch.uint8)[bws - 4096:bws].cpu().numpy().view(np.uint64).reshape(8, 8, 8)   # [block][wave][seg]
names = ["max+split", "barrier1", "T2 mma", "barrier2", "T1 mma", "barrier3", "record wait", "gate math"]
per_step = raw.astype(np.float64) / a.seq_len
print("cycles per step (mean over 8 blocks), per wave:")
for w in range(8):
    v = per_step[:, w, :len(names)].mean(0)
    if v.sum() > 0:
        print("wave", w, " ".join("%7.0f" % q for q in v), " total %.0f" % v.sum())
print("segments:", names)
