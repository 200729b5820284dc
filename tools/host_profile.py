#!/usr/bin/env python3
"""Developer tool: where the HOST time of one forward call goes (cProfile over N no-grad calls of the cfg2 module; the GPU work is
queued asynchronously, so cumulative times are Python / driver-call time, not kernel time)."""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorized-rnn_amd"))
import torch
from tensorized_rnn.tt_lstm import TTLSTM
dev = torch.device("cuda:0")
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    m = TTLSTM(1, 256, 1, dev, n_cores=3, tt_rank=8).eval()
B, T = int(os.environ.get("HP_B", "64")), int(os.environ.get("HP_T", "8"))      # short sequences: the host must not wait for the GPU
x = torch.rand(B, T, 1, device=dev)
N = 300
with torch.no_grad():
    for _ in range(20):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("host time per call (queueing only): %.1f us" % ((t1 - t0) / N * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        m(x)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
