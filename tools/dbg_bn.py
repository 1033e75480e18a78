import sys, ctypes
sys.path.insert(0,'/root/repo/fa-vae_amd')
import numpy as np, torch, torch.nn.functional as F
from favae_hip import lib as L
raw_=L.raw()
torch.manual_seed(0)
R,C=int(sys.argv[1]),int(sys.argv[2])
x=torch.randn(R,C,dtype=torch.float64)*2+0.5
gamma=torch.rand(C,dtype=torch.float64)+0.5; beta=torch.randn(C,dtype=torch.float64)
x.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True)
y=F.batch_norm(x.t().reshape(1,C,R).permute(0,1,2).contiguous().view(1,C,R,1).permute(2,1,0,3).contiguous().view(R,C,1,1),None,None,gamma,beta,True,0.1,1e-5)
a=F.leaky_relu(y,0.01)
da=torch.randn(R,C,1,1,dtype=torch.float64)
a.backward(da)
dev='cuda'
xd=x.detach().float().to(dev).contiguous()
mean=torch.zeros(C,device=dev); invstd=torch.zeros(C,device=dev)
scratch=torch.zeros(2*1024*64+4096,dtype=torch.float64,device=dev)
i=ctypes.c_int; ll=ctypes.c_longlong; f=ctypes.c_float; sz=ctypes.c_size_t
L.check(raw_.favae_bn_stats(L.ptr(xd), ll(R), i(C), f(1e-5), f(0.1), L.ptr(mean), L.ptr(invstd), None, None, L.ptr(scratch), sz(scratch.numel()), L.stream_ptr()))
act=torch.zeros(R,C,device=dev)
g32=gamma.detach().float().to(dev); b32=beta.detach().float().to(dev)
L.check(raw_.favae_bn_lrelu_fwd(L.ptr(xd),L.ptr(mean),L.ptr(invstd),L.ptr(g32),L.ptr(b32),L.ptr(act),ll(R),i(C),i(1),i(0),L.stream_ptr()))
print('act err', (act.cpu().double()-a.detach().view(R,C)).abs().max().item())
dad=da.view(R,C).float().to(dev).contiguous()
draw=torch.zeros(R,C,device=dev); dg=torch.zeros(C,device=dev); db=torch.zeros(C,device=dev); dbias=torch.zeros(C,device=dev); s1s2=torch.zeros(2*C,device=dev)
L.check(raw_.favae_bn_lrelu_bwd(L.ptr(dad),L.ptr(act),L.ptr(xd),L.ptr(mean),L.ptr(invstd),L.ptr(g32),L.ptr(draw),L.ptr(dg),L.ptr(db),L.ptr(dbias),ll(R),i(C),L.ptr(scratch),sz(scratch.numel()),L.ptr(s1s2),i(0),L.stream_ptr()))
torch.cuda.synchronize()
print('dgamma err', (dg.cpu().double()-gamma.grad).abs().max().item(), gamma.grad.abs().max().item())
print('dbeta err', (db.cpu().double()-beta.grad).abs().max().item(), beta.grad.abs().max().item())
print('draw err', (draw.cpu().double()-x.grad).abs().max().item(), x.grad.abs().max().item())
print('dbias', dbias.abs().max().item())
e=(draw.cpu().double()-x.grad).abs()
print('bad rows', (e.max(1).values>1e-4).nonzero().flatten()[:20], 'bad cols', (e.max(0).values>1e-4).nonzero().flatten()[:20])
