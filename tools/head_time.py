import sys, os, contextlib, io
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tensorized-rnn_amd"))
import torch
from t3nsor.layers import TTLinear
from ttrnn_hip import functional as F
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    lin = TTLinear(in_features=1024, out_features=256, bias=True, auto_shapes=True, d=4, tt_rank=32).to(dev)
cores = list(lin.weight_t.tt_cores)
spec = F.TTSpec.from_cores(cores)
packed = spec.pack(cores)
x = torch.randn(128, 1024, device=dev); dy = torch.randn(128, 256, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("dx only   %.3f ms" % t(lambda: F._ttlinear_backward(spec, packed, x, dy, True, False, False)))
print("dw only   %.3f ms" % t(lambda: F._ttlinear_backward(spec, packed, x, dy, False, True, False)))
print("dw + bias %.3f ms" % t(lambda: F._ttlinear_backward(spec, packed, x, dy, False, True, True)))
print("all       %.3f ms" % t(lambda: F._ttlinear_backward(spec, packed, x, dy, True, True, True)))
