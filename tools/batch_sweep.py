import sys, torch, time, contextlib, io
sys.path.insert(0,"tensorized-rnn_amd"); sys.path.insert(0,"examples")
from models import MNISTClassifier
import torch.nn.functional as F
dev=torch.device("cuda")
for (H,d,r,gru,inp,T) in ((256,3,8,False,1,784),(256,3,16,False,40,160),(256,3,8,True,1,784)):
    with contextlib.redirect_stdout(io.StringIO()):
        m=MNISTClassifier(inp,10,H,1,dev,gru=gru,n_cores=d,tt_rank=r).to(dev)
    opt=torch.optim.Adam(m.parameters(),lr=1e-3)
    for B in (1,7,64,128,129,256,257,512,1024,2048):
        x=torch.rand(B,T,inp,device=dev); y=torch.randint(0,10,(B,),device=dev)
        def ev():
            with torch.no_grad(): m(x)
        def tr():
            opt.zero_grad(); F.nll_loss(m(x),y).backward(); opt.step()
        out=[]
        for fn in (ev,tr):
            fn(); torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(3): fn()
            torch.cuda.synchronize(); out.append((time.perf_counter()-t0)/3*1e3)
        print("H%d r%d %s in%d T%d B %5d eval %7.3f ms train %7.3f ms  (per 256 samples: %.3f / %.3f)"%(H,r,"gru" if gru else "lstm",inp,T,B,out[0],out[1],out[0]*256/max(B,256),out[1]*256/max(B,256)), flush=True)
