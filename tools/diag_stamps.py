#!/usr/bin/env python3
"""Developer diagnostic: per-phase cycle shares of the fused-core forward kernels (TTRNN_DIAG=1 build variant with
s_memtime stamps): the TT-LSTM kernels, with DIAG_CELL=gru the fp32 TT-GRU kernel k_gru_fwd_f10vh, with DIAG_CELL=naive k_rnn_fwd_f10n.
Shares only — never quote the diagnostic build's run time."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
os.environ["TTRNN_DIAG"] = "1"
import numpy as np, torch
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU
from ttrnn_hip import functional as F

dev = torch.device("cuda:0")
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    RANK, INP = int(os.environ.get("DIAG_RANK", "8")), int(os.environ.get("DIAG_IN", "1"))
    GRU = os.environ.get("DIAG_CELL", "lstm") == "gru"
    NAIVE = os.environ.get("DIAG_CELL", "lstm") == "naive"      # k_rnn_fwd_f10n (ttrnn_fast_f10n.hip): the naive per-gate TT-LSTM
    HID = int(os.environ.get("DIAG_H", "256"))      # 512 / 384: the eight- / six-wave instantiations of k_lstm_fwd_f10q
    m = (TTGRU if GRU else TTLSTM)(INP, HID, 1, dev, n_cores=3, tt_rank=RANK, is_naive=NAIVE)
B, T = int(os.environ.get("DIAG_B", "64")), int(os.environ.get("DIAG_T", "784"))
x = torch.rand(B, T, INP, device=dev, requires_grad=True)     # requires_grad -> reserve buffer exists
captured = {}
orig = F._TTRnnLayerFn.forward
def fwd(ctx, *a):
    osave = ctx.save_for_backward
    def save(*ts):
        captured["reserve"] = ts[2]
        return osave(*ts)
    ctx.save_for_backward = save
    return orig(ctx, *a)
F._TTRnnLayerFn.forward = staticmethod(fwd)
for _ in range(2):
    out, _ = m(x)
torch.cuda.synchronize()
raw = captured["reserve"][:8 * 8 * 8 * 2].cpu().numpy().view(np.uint64).reshape(8, 8, 8)   # [block][wave][seg]
import ttrnn_hip
NW = 8 if (os.environ.get("TTRNN_F10_NB1") == "1" or int(os.environ.get("DIAG_H", "256")) > 256) else 4      # default: the four-wave kernel k_lstm_fwd_f10q (waves 4..7: unused slots)
if GRU or NAIVE:                             # k_gru_fwd_f10vh (ttrnn_fast_f10gh.hip) / k_rnn_fwd_f10n (ttrnn_fast_f10n.hip)
    names = ["S10 mma+gbuf", "barrier1", "gates", "S2+split", "barrier2", "-", "-", "-"]
elif ttrnn_hip.get_fp32_math() == "split" and (int(os.environ.get("TTRNN_DEV", "0")) & 512) and RANK == 8 \
        and os.environ.get("TTRNN_F10_NB1") != "1":      # TTRNN_DEV=512: k_lstm_fwd_f10s (ttrnn_fast_f10s.hip, A/B kernel)
    names = ["S10 mma", "gates", "S2+split", "stores", "barrier", "-", "-", "-"]
elif ttrnn_hip.get_fp32_math() == "split":     # k_lstm_fwd_f10q / k_lstm_fwd_f10 (ttrnn_fast_f10q.hip, ttrnn_fast_f10.hip)
    names = ["S2+split", "barrier1", "S10 mma", "gates+stores", "barrier2", "-", "-", "-"]
else:                                        # k_lstm_fwd_fused (ttrnn_fast.hip)
    names = ["S2 mma+store", "barrier1", "S1 mma+store", "barrier2", "S0 mma", "gates", "barrier3", "-"]
per_step = raw.astype(np.float64) / T
print("cycles per step (mean over 8 blocks), per wave:")
for w in range(NW):
    print("wave", w, " ".join("%7.0f" % v for v in per_step[:, w, :7].mean(0)), " total %.0f" % per_step[:, w, :7].mean(0).sum())
print("segments:", names[:7])
