"""where does the TT-GRU training harness' occasional 20 ms iteration come from?  per-iteration wall time + allocator counters"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tensorized-rnn_amd"), os.path.join(ROOT, "examples")]
import numpy as np, torch, torch.nn.functional as F
from models import MNISTClassifier
dev = torch.device("cuda")
gru = "--lstm" not in sys.argv
with contextlib.redirect_stdout(io.StringIO()):
    model = MNISTClassifier(40, 256, 768, 1, dev, gru=gru, n_cores=2, tt_rank=2).to(dev)
data = torch.rand(512, 160, 40, device=dev)
target = torch.randint(0, 256, (512,), device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    loss = F.nll_loss(model(data), target)
    loss.backward()
    opt.step()
def stats():
    s = torch.cuda.memory_stats()
    return s["num_device_alloc"], s["num_device_free"], s["num_alloc_retries"], s["reserved_bytes.all.current"] >> 20
step(); torch.cuda.synchronize()
prev = stats()
for i in range(12):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cur = stats()
    print("iter %2d: %.3f ms   device allocs +%d frees +%d retries +%d reserved %d MB" % (i, dt * 1e3, cur[0] - prev[0], cur[1] - prev[1], cur[2] - prev[2], cur[3]))
    prev = cur
