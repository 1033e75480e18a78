import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/fa-vae_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from conftest import img_inputs
from oracle import vae_torch as vt
from favae_hip.engine import VaeEngine
z,Bn=20,16
torch.manual_seed(0); np.random.seed(0)
P,Bf = vt.init_state(z,3)
names=list(P.keys())
eng = VaeEngine(z,3,dtype=sys.argv[1] if len(sys.argv)>1 else 'f32')
sd={k:v.detach().clone() for k,v in P.items()}; sd.update({k:v.clone() for k,v in Bf.items()})
eng.load_state_dict(sd)
img,_=img_inputs(Bn)
prior=torch.from_numpy(np.random.RandomState(7).randn(Bn,z)*0.5)
tau=torch.Tensor(np.asarray(np.array([1.7])))
images=torch.tensor(img)
data=eng.upload_images(img)
idx=torch.arange(Bn,dtype=torch.int32,device='cuda')
eps=torch.randn(Bn,z,generator=torch.Generator().manual_seed(100))
loss,rec_err,kl,mu,std,zz,rec = vt.loss_terms(P,Bf,images,prior,tau,0.01,1,eps=eps)
loss.backward()
# float64 oracle
P64={k:v.detach().double().requires_grad_(True) for k,v in P.items()}
B64={k:(v.clone().double() if v.dtype.is_floating_point else v.clone()) for k,v in sd.items() if k not in P}
l64,*_ = vt.loss_terms(P64,B64,images.double(),prior,tau.double(),0.01,1,eps=eps.double())
l64.backward()
m=eng.train_fwdbwd(data,idx,Bn,eps.cuda(),prior.cuda(),float(tau.item()),1.0,0.01).cpu().numpy()
print('loss', m, rec_err.item(), kl.item())
g=eng.unpack(eng.G)
def rel(a,b): return float(np.abs(a-b).max()/(np.abs(b).max()+1e-30))
def cos(a,b):
    a=a.ravel().astype(np.float64); b=b.ravel().astype(np.float64); return float(a@b/(np.linalg.norm(a)*np.linalg.norm(b)+1e-300))
for k in names:
    t=P64[k].grad.numpy()
    if 'bias' in k and ('.0.' in k) and k!='final_layer.3.bias': continue
    print('cos %.5f  ' % cos(g[k].numpy(), t), end='')
    print('%-28s hip-vs-f64 %.2e  torch32-vs-f64 %.2e  hip-vs-torch32 %.2e  |g|max %.2e' % (k, rel(g[k].numpy().astype(np.float64),t), rel(P[k].grad.numpy().astype(np.float64),t), rel(g[k].numpy(),P[k].grad.numpy()), np.abs(t).max()))
