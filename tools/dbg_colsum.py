import sys, ctypes
sys.path.insert(0,'/root/repo/fa-vae_amd')
import numpy as np, torch
from favae_hip import lib as L
raw_=L.raw()
i=ctypes.c_int; ll=ctypes.c_longlong; sz=ctypes.c_size_t
dev='cuda'
R,C=65536,64
scratch=torch.zeros(2*1024*64+4096,dtype=torch.float64,device=dev)
xs=[torch.randn(R,C,device=dev)*(k+1) for k in range(4)]
ref=[x.double().sum(0).float() for x in xs]
out=torch.zeros(C,device=dev)
bad=0
for it in range(400):
    k=it%4
    L.check(raw_.favae_colsum(L.ptr(xs[k]), ll(R), i(C), L.ptr(out), L.ptr(scratch), sz(scratch.numel()), i(0), L.stream_ptr()))
    e=(out-ref[k]).abs().max().item()
    if e>1e-2:
        bad+=1
        if bad<5: print('iter',it,'err',e, 'nbadch', ((out-ref[k]).abs()>1e-2).sum().item())
print('bad',bad,'of 400')
