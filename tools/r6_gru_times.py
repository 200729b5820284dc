import sys, os, time, contextlib, io, torch, numpy as np
ROOT=os.getcwd(); sys.path[:0]=[ROOT, ROOT+'/tensorized-rnn_amd', ROOT+'/examples']
from models import MNISTClassifier
import torch.nn.functional as F
dev=torch.device('cuda')
with contextlib.redirect_stdout(io.StringIO()):
    m=MNISTClassifier(40,256,768,1,dev,gru=True,n_cores=2,tt_rank=2).to(dev)
x=torch.rand(512,160,40,device=dev); t=torch.randint(0,256,(512,),device=dev)
opt=torch.optim.Adam(m.parameters(),lr=1e-3)
for i in range(12):
    torch.cuda.synchronize(); t0=time.perf_counter()
    opt.zero_grad(); l=F.nll_loss(m(x),t); l.backward(); opt.step()
    torch.cuda.synchronize(); print(i, '%.3f ms'%((time.perf_counter()-t0)*1e3), float(l))
