import sys, os, contextlib, io
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "tensorized-rnn_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import torch, ttrnn_hip
from golden_io import build_module
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = build_module(dict(kind="ttgru", input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8), dev).to(torch.bfloat16)
for (B, T, with_h0) in ((5, 37, False), (64, 100, True), (256, 784, False)):
    x = torch.rand(B, T, 1, device=dev).to(torch.bfloat16)
    h0 = (torch.randn(B, 256, device=dev) * 0.5).to(torch.bfloat16) if with_h0 else None
    with torch.no_grad():
        o0, h_0 = m(x, h0)
        with ttrnn_hip.option("dev", 32):      # the eight-wave kernel
            o1, h_1 = m(x, h0)
    torch.cuda.synchronize()
    print(B, T, with_h0, "bitwise equal:", bool(torch.equal(o0, o1) and torch.equal(h_0, h_1)), float((o0.float() - o1.float()).abs().max()))
# training forward (reserve) too
x = torch.rand(7, 19, 1, device=dev).to(torch.bfloat16)
gs = []
for d in (0, 32):
    m.zero_grad()
    with ttrnn_hip.option("dev", d):
        o, h = m(x)
        o.float().sum().backward()
    gs.append([p.grad.clone() for p in m.parameters()])
print("grads equal:", all(torch.equal(a, b) for a, b in zip(*gs)))
