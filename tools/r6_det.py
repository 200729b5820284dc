import sys, os, torch, contextlib, io
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'tensorized-rnn_amd'), os.path.join(os.getcwd(),'tests')]
from golden_io import build_module
from ttrnn_hip import functional as F
import ttrnn_hip
dev=torch.device('cuda:0')
torch.manual_seed(41)
m=build_module(dict(kind="ttlstm", input_size=40, hidden_size=768, num_layers=1, n_cores=2, tt_rank=2), dev)
g=torch.Generator().manual_seed(43)
B,T=5,9
x=torch.randn(B,T,40,generator=g); w=torch.randn(B,T,768,generator=g)
def run():
    m.zero_grad()
    F.DEBUG_BWD_STATS=[]
    xg=x.to(dev).requires_grad_(True)
    out,(hT,cT)=m(xg)
    (out*w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    st=F.DEBUG_BWD_STATS[0]
    F.DEBUG_BWD_STATS=None
    return out.detach().clone(), st[2].clone(), st[1].clone() if st[1] is not None else None, {n:p.grad.clone() for n,p in m.named_parameters()}, xg.grad.clone()
a=run(); b=run()
print("out", torch.equal(a[0],b[0]), "dgates", torch.equal(a[1],b[1]), "stats", None if a[2] is None else torch.equal(a[2],b[2]))
for n in a[3]: print(n, torch.equal(a[3][n], b[3][n]), float((a[3][n]-b[3][n]).abs().max()))
print("dx", torch.equal(a[4],b[4]))
