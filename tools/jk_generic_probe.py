import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
import cgc_net_amd
from cgc_net_amd import kernels
from oracle.flat_ref import TorchKernels
REF=TorchKernels(); K=kernels.get(); DEV='cuda:0'
for C,n in ((24,64),(24,300),(22,300),(26,300),(32,300),(24,1300)):
    H=3*C//2
    torch.manual_seed(C+n)
    lstm_mod=torch.nn.LSTM(C,H,bidirectional=True,batch_first=True); att=torch.nn.Linear(2*H,1)
    xs=torch.randn(n,3*C); p=lstm_mod
    lstm=[t.detach() for t in (p.weight_ih_l0,p.weight_hh_l0,p.bias_ih_l0,p.bias_hh_l0,p.weight_ih_l0_reverse,p.weight_hh_l0_reverse,p.bias_ih_l0_reverse,p.bias_hh_l0_reverse)]
    w_att,b_att=att.weight.detach().reshape(-1),att.bias.detach()
    npad=-(-n//1024)*1024
    res={}
    for name,K_,dev in (('ref',REF,'cpu'),('hip',K,DEV)):
        t=lambda v:v.to(dev)
        out=torch.empty(n,C,device=dev); HS,CS=torch.zeros(6*H,npad,device=dev),torch.zeros(6*H,npad,device=dev)
        K_.jk_fwd(t(xs),n,npad,C,[t(v) for v in lstm],t(w_att),t(b_att),out,HS,CS)
        dout=t(torch.randn(n,C,generator=torch.Generator().manual_seed(3)))
        dxs=torch.empty(n,3*C,device=dev)
        DGT=torch.full((2,4*H+1,3*npad),7.0,device=dev); INT=torch.full((2,C+2*H+1,3*npad),7.0,device=dev); DHC=torch.empty(2,2,H,npad,device=dev)
        K_.jk_bwd(t(xs),dout,n,npad,C,[t(v) for v in lstm],t(w_att),t(b_att),HS,CS,dxs,DGT,INT,DHC)
        res[name]=dict(out=out.cpu(),dxs=dxs.cpu(),DGT=DGT[:,:,:].cpu(),INT=INT.cpu())
    e=(res['hip']['dxs']-res['ref']['dxs']).abs()
    print(C,n,'dxs err max',float(e.max()),'per t',[float(e[:,t*C:(t+1)*C].max()) for t in range(3)],'rows bad',int((e.max(1)[0]>1e-4).sum()))
    eg=(res['hip']['DGT']-res['ref']['DGT']).abs()
    print('   DGT err per dir/gate-row-block', [[float(eg[d,g*H:(g+1)*H].max()) for g in range(4)] for d in range(2)], 'score row', float(eg[:,4*H].max()))
    ei=(res['hip']['INT']-res['ref']['INT']).abs(); print('   INT err', float(ei.max()))
