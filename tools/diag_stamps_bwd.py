#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the fused-core reverse-time kernel k_lstm_bwd_f10 (TTRNN_DIAG=1
variant with s_memtime stamps; the stamps land behind the fragments in the backward workspace).  Shares only — never
quote the diagnostic build's run time."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
os.environ["TTRNN_DIAG"] = "1"
import numpy as np, torch
from tensorized_rnn.tt_lstm import TTLSTM
from ttrnn_hip import functional as F

dev = torch.device("cuda:0")
torch.manual_seed(1111)
RANK = int(os.environ.get("DIAG_RANK", "8"))
with contextlib.redirect_stdout(io.StringIO()):
    m = TTLSTM(1, 256, 1, dev, n_cores=3, tt_rank=RANK)
B, T = int(os.environ.get("DIAG_B", "64")), int(os.environ.get("DIAG_T", "784"))
x = torch.rand(B, T, 1, device=dev, requires_grad=True)
kept = []
orig_ws = F._workspace
def ws(nbytes, device):
    t = orig_ws(nbytes, device)
    kept.append(t)
    return t
F._workspace = ws
out, _ = m(x)
kept.clear()
out.square().sum().backward()
torch.cuda.synchronize()
bws = kept[0]                                   # first workspace of backward() = ttrnn_rnn_backward's
raw = bws[-4096:].cpu().numpy().view(np.uint64).reshape(8, 8, 8)   # [block][wave][seg]
names = ["G gate grads", "barrier1", "T01 + split", "barrier2", "T2", "barrier3"]
per_step = raw.astype(np.float64) / T
print("cycles per step (mean over 8 blocks), per wave:")
for w in range(8):
    print("wave", w, " ".join("%7.0f" % v for v in per_step[:, w, :6].mean(0)), " total %.0f" % per_step[:, w, :6].mean(0).sum())
print("segments:", names)
