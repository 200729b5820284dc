#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the fused-core reverse-time LSTM kernel k_lstm_bwd_f10 (TTRNN_DIAG=1 build
variant with s_memtime stamps in the 4 KB behind its fragments in the workspace).  Shares only — never quote the diagnostic
build's run time.   DIAG_RANK=8|16 DIAG_IN=1 DIAG_B=64 DIAG_T=784 python tools/diag_stamps_bwd.py"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
os.environ["TTRNN_DIAG"] = "1"
import ctypes
import numpy as np, torch
from tensorized_rnn.tt_lstm import TTLSTM
from ttrnn_hip import functional as F, _lib

dev = torch.device("cuda:0")
torch.manual_seed(1111)
RANK, INP = int(os.environ.get("DIAG_RANK", "8")), int(os.environ.get("DIAG_IN", "1"))
HID = int(os.environ.get("DIAG_H", "256"))      # 512: the reference's default benchmark shape (rank 8 only)
with contextlib.redirect_stdout(io.StringIO()):
    m = TTLSTM(INP, HID, 1, dev, n_cores=3, tt_rank=RANK)
B, T = int(os.environ.get("DIAG_B", "64")), int(os.environ.get("DIAG_T", "784"))
x = torch.rand(B, T, INP, device=dev)
seen = []
orig_ws = F._workspace
def ws(nbytes, device):
    t = orig_ws(nbytes, device)
    seen.append((nbytes, t))
    return t
F._workspace = ws
desc = m._all_layers[0]._layer_spec().desc(B, T, _lib.TTRNN_F32)
bws = _lib.load().ttrnn_rnn_backward_workspace(ctypes.byref(desc))
for _ in range(2):
    seen.clear()
    out, _ = m(x)
    out.sum().backward()
torch.cuda.synchronize()
buf = next(t for n, t in seen if n == bws)
FT, NM1, NM2 = (16, 2, 4) if RANK == 8 else (32, 2, 8)
off = (FT * NM1 + NM2) * 3 * 64 * 16
if HID == 512:      # two-piece kernel alone: [header | fragments | stamps] (launch_rnn_bwd_f10_h512)
    off = 1088 * 4 + (32 * 4 + 4) * 2 * 64 * 16
raw = buf.view(torch.uint8)[off:off + 8 * 8 * 8 * 8].cpu().numpy().view(np.uint64).reshape(8, 8, 8)   # [block][wave][seg]
# TTRNN_GEMM_PIECES=3: the three-bf16-piece kernel (6 segments); default: the two-piece fp16 kernel (8 segments)
if os.environ.get("TTRNN_GEMM_PIECES") == "3":
    names = ["G gates", "barrier1", "T01 mma+split", "barrier2", "T2 mma", "barrier3"]
elif raw[:, :4, 6:].sum() > 0:      # k_lstm_bwd_f10h: eight waves, four barriers (one sample per CU at H = 256; dev bit 15)
    names = ["G gates", "barrier1", "split", "barrier1b", "T01 mma", "barrier2", "T2 mma", "barrier3"]
else:      # k_lstm_bwd_f10l: one wave per 64 units (waves 4-7 exist at H = 512 only), T2 wave-local
    names = ["G gates", "barrier1", "split", "barrier2", "T01 mma", "T2 mma"]
per_step = raw.astype(np.float64) / T
print("cycles per step (mean over 8 blocks), per wave:")
for w in range(8):
    v = per_step[:, w, :len(names)].mean(0)
    print("wave", w, " ".join("%7.0f" % a for a in v), " total %.0f" % v.sum())
print("segments:", names)
