#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the runtime-shape forward kernel k_g2_fwd (TTRNN_DIAG=1 instantiation: one unit per
thread, head fragments resident in eight slots, input_size != 1; its stamps land at the head of the training reserve).  Shares only.
   python tools/diag_stamps_g2fwd.py [--gru] [--naive_tt] [--hidden_size H] [--ncores d] [--ttrank r] [--in_size n] [--batch_size B] [--seq_len T]"""
import argparse, contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TTRNN_DIAG"] = "1"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import numpy as np, torch
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU
from ttrnn_hip import functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--gru", action="store_true"); ap.add_argument("--naive_tt", action="store_true")
ap.add_argument("--in_size", type=int, default=40); ap.add_argument("--hidden_size", type=int, default=384)
ap.add_argument("--ncores", type=int, default=3); ap.add_argument("--ttrank", type=int, default=8)
ap.add_argument("--batch_size", type=int, default=64); ap.add_argument("--seq_len", type=int, default=160)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    m = (TTGRU if a.gru else TTLSTM)(a.in_size, a.hidden_size, 1, dev, n_cores=a.ncores, tt_rank=a.ttrank, is_naive=a.naive_tt)
x = torch.rand(a.batch_size, a.seq_len, a.in_size, device=dev)
print("forward route:", F.rnn_route(m._all_layers[0]._layer_spec(), a.batch_size, a.seq_len), "| samples per workgroup:",
      F.rnn_samples_per_workgroup(m._all_layers[0]._layer_spec(), a.batch_size, a.seq_len))
seen = []
orig = F._alloc
def alloc(shape, dtype, device):
    t = orig(shape, dtype, device)
    seen.append(t)
    return t
F._alloc = alloc
for _ in range(2):
    seen.clear()
    out = m(x)[0]
torch.cuda.synchronize()
H, B, T = a.hidden_size, a.batch_size, a.seq_len
res = max((t for t in seen if t.dtype == torch.float32 and t.dim() == 1), key=lambda t: t.numel())      # the reserve: the largest flat fp32 buffer
raw = res.view(torch.uint8)[:4 * 8 * 8 * 8].cpu().numpy().view(np.uint64).reshape(4, 8, 8)   # [block][wave][seg]
names = ["stage 1 + split", "barrier1", "stage 2", "barrier2", "gates + stores", "barrier3"]
per_step = raw.astype(np.float64) / T
print("cycles per step (mean over 4 blocks), per wave:")
for w in range(8):
    v = per_step[:, w, :len(names)].mean(0)
    if 0 < v.sum() < 1e9:
        print("wave", w, " ".join("%7.0f" % q for q in v), " total %.0f" % v.sum())
print("segments:", names)
