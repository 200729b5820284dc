"""Are the parameter gradients of two identical backward passes bitwise equal?  (tools: which tensors differ)"""
import sys, torch, contextlib, io
sys.path.insert(0,"tensorized-rnn_amd"); sys.path.insert(0,"examples"); sys.path.insert(0,"tests")
from golden_io import build_module
dev=torch.device("cuda")
cases=[dict(kind="ttlstm", input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8),
       dict(kind="ttlstm", input_size=40, hidden_size=256, num_layers=3, n_cores=3, tt_rank=16),
       dict(kind="ttgru", input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8),
       dict(kind="ttlstm", input_size=256, hidden_size=512, num_layers=1, n_cores=3, tt_rank=8),
       dict(kind="ttlstm", input_size=1, hidden_size=1024, num_layers=1, n_cores=2, tt_rank=16),
       dict(kind="ttlstm", input_size=1, hidden_size=256, num_layers=1, n_cores=3, tt_rank=8, is_naive=True)]
for meta in cases:
    torch.manual_seed(3)
    m=build_module(meta, dev)
    B,T=(96,64)
    x=torch.randn(B,T,meta["input_size"],device=dev); w=torch.randn(B,T,meta["hidden_size"],device=dev)
    gs=[]
    for rep in range(3):
        m.zero_grad()
        out=m(x)[0]; (out*w).sum().backward()
        gs.append([p.grad.clone() for p in m.parameters()])
    same=all(torch.equal(a,b) for a,b in zip(gs[0],gs[1])) and all(torch.equal(a,b) for a,b in zip(gs[0],gs[2]))
    names=[n for n,_ in m.named_parameters()]
    diff=[n for n,a,b,c in zip(names,gs[0],gs[1],gs[2]) if not (torch.equal(a,b) and torch.equal(a,c))]
    print("   differing:", diff[:8], "of", len(names))
    print({k:v for k,v in meta.items() if k in ("kind","input_size","hidden_size","num_layers","n_cores","tt_rank","is_naive")}, "gradients bitwise repeatable:", same)
