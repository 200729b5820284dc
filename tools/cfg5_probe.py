import sys, time, contextlib, io
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tensorized-rnn_amd"))
import torch
from tensorized_rnn.tt_lstm import TTLSTM
dev = torch.device('cuda:0')
torch.manual_seed(1111)
with contextlib.redirect_stdout(io.StringIO()):
    m = TTLSTM(1024, 1024, 1, dev, n_cores=4, tt_rank=32)
for B, T in ((16, 4), (128, 8)):
    x = torch.rand(B, T, 1024, device=dev)
    with torch.no_grad():
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("cfg5 generic B=%d T=%d: %.1f ms  (%.2f ms/step, %.2f TFLOP/s)" % (B, T, dt*1e3, dt*1e3/T, 70267904.0*B*T/dt/1e12))
