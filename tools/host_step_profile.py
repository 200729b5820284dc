#!/usr/bin/env python3
"""Developer tool: HOST time of one training step of the reference's benchmark harness (examples/benchmarking.py semantics) at the
speaker-encoder size with a short sequence — the GPU work is tiny then, so wall time per step is Python / driver-call time:
    python tools/host_step_profile.py [--gru]"""
import contextlib, cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tensorized-rnn_amd"), os.path.join(ROOT, "examples")]
import numpy as np
import torch
import torch.nn.functional as F
from models import MNISTClassifier
dev = torch.device("cuda:0")
gru = "--gru" in sys.argv
with contextlib.redirect_stdout(io.StringIO()):
    model = MNISTClassifier(40, 256, 768, 1, dev, gru=gru, n_cores=2, tt_rank=2).to(dev)
B, T = 512, 4
x = torch.rand(B, T, 40, device=dev)
target = torch.from_numpy(np.random.randint(0, 256, B).astype("int64")).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    out = model(x)
    loss = F.nll_loss(out, target)
    loss.backward()
    opt.step()
for _ in range(10):
    step()
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step()
    torch.cuda.synchronize()
t1 = time.perf_counter()
print("wall time per step with a 4-step sequence (host bound): %.1f us" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:40]))
