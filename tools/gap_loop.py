#!/usr/bin/env python3
"""Driver for tools/gap_report.sh: N no-grad forwards of the cfg2 module (GAP_NOOUT=1: need_outputs=False; GAP_PREP=1: prepared)."""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import torch
from tensorized_rnn.tt_lstm import TTLSTM
from tensorized_rnn.gru import TTGRU
dev = torch.device("cuda:0")
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    m = (TTGRU if os.environ.get("GAP_GRU") else TTLSTM)(1, 256, 1, dev, n_cores=3, tt_rank=8).eval()
x = torch.rand(64, 784, 1, device=dev)
if os.environ.get("GAP_PREP"):
    m.prepare_for_inference()
kw = dict(need_outputs=False) if os.environ.get("GAP_NOOUT") else {}
with torch.no_grad():
    for _ in range(25):
        r = m(x, **kw)
torch.cuda.synchronize()
