import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/fa-vae_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, torch.nn.functional as F
from conftest import img_inputs
from oracle import vae_torch as vt
from favae_hip.engine import VaeEngine
z,Bn=20,16
torch.manual_seed(0); np.random.seed(0)
P,Bf = vt.init_state(z,3)
eng = VaeEngine(z,3,dtype='f32')
sd={k:v.detach().clone() for k,v in P.items()}; sd.update({k:v.clone() for k,v in Bf.items()})
eng.load_state_dict(sd)
img,_=img_inputs(Bn)
prior=torch.from_numpy(np.random.RandomState(7).randn(Bn,z)*0.5)
tau=torch.Tensor(np.asarray(np.array([1.7])))
images=torch.tensor(img)
data=eng.upload_images(img)
idx=torch.arange(Bn,dtype=torch.int32,device='cuda')
eps=torch.randn(Bn,z,generator=torch.Generator().manual_seed(100))
# manual forward with retained intermediates (float64 for truth)
def run(dt):
    Pd={k:v.detach().to(dt).requires_grad_(True) for k,v in P.items()}
    keep={}
    x=images.to(dt)
    def blk(x,pre,tr,name):
        if tr: r=F.conv_transpose2d(x,Pd[pre+'.0.weight'],Pd[pre+'.0.bias'],stride=2,padding=1,output_padding=1)
        else: r=F.conv2d(x,Pd[pre+'.0.weight'],Pd[pre+'.0.bias'],stride=2,padding=1)
        r.retain_grad(); keep['raw_'+name]=r
        y=F.batch_norm(r,None,None,Pd[pre+'.1.weight'],Pd[pre+'.1.bias'],True,0.1,1e-5)
        a=F.leaky_relu(y,0.01); a.retain_grad(); keep['act_'+name]=a
        return a
    for i in range(5): x=blk(x,'enc.%d'%i,False,str(i))
    f=torch.flatten(x,1)
    mu=F.linear(f,Pd['fc_mu.weight'],Pd['fc_mu.bias']); lv=F.linear(f,Pd['fc_var.weight'],Pd['fc_var.bias'])+1e-6
    std=torch.exp(0.5*lv); zz=eps.to(dt)*std+mu
    d=F.linear(zz,Pd['decoder_input.weight'],Pd['decoder_input.bias']); d.retain_grad(); keep['raw_6']=d
    x=d.view(-1,1024,2,2)
    for i in range(4): x=blk(x,'dec.%d'%i,True,str(7+i))
    x=blk(x,'final_layer',True,'11')
    r=F.conv2d(x,Pd['final_layer.3.weight'],Pd['final_layer.3.bias'],padding=1); r.retain_grad(); keep['raw_12']=r
    rec=torch.sigmoid(r)
    rec_err=-vt.gaussian_ll(images.to(dt),rec,0.01)
    stdP=torch.sqrt(torch.ones_like(prior)*(1/tau.double()))
    kl=vt.kl_div(mu,std,prior,stdP)
    (rec_err+kl).backward()
    return keep
k64=run(torch.float64); k32=run(torch.float32)
m=eng.train_fwdbwd(data,idx,Bn,eps.cuda(),prior.cuda(),float(tau.item()),1.0,0.01).cpu().numpy()
def rel(a,b): return float(np.abs(a-b).max()/(np.abs(b).max()+1e-30))
def nhwc(t): return t.detach().permute(0,2,3,1).contiguous().numpy()
for l in [12,11,10,9,8,7,4,3,2,1,0]:
    name=str(l)
    t64=k64['raw_'+name]; shp=nhwc(t64).shape
    raw=eng.debug(0,l,shp,torch.float32).cpu().numpy()
    draw=eng.debug(3,l,shp,torch.float32).cpu().numpy()
    print('L%-2d raw  hip %.2e torch32 %.2e | draw hip %.2e torch32 %.2e' % (l, rel(raw,nhwc(t64)), rel(nhwc(k32['raw_'+name]),nhwc(t64)), rel(draw,nhwc(t64.grad)), rel(nhwc(k32['raw_'+name].grad),nhwc(t64.grad))), end='')
    if 'act_'+name in k64:
        act=eng.debug(1,l,shp,torch.float32).cpu().numpy(); dact=eng.debug(2,l,shp,torch.float32).cpu().numpy()
        print(' | act hip %.2e | dact hip %.2e torch32 %.2e' % (rel(act,nhwc(k64['act_'+name])), rel(dact,nhwc(k64['act_'+name].grad)), rel(nhwc(k32['act_'+name].grad),nhwc(k64['act_'+name].grad))))
    else: print()
t64=k64['raw_11']; shp=nhwc(t64).shape
draw=eng.debug(3,11,shp,torch.float32).cpu().numpy()
e=np.abs(draw-nhwc(t64.grad))
print('err by n', e.max(axis=(1,2,3)))
print('err by c', e.max(axis=(0,1,2))[:16])
print('err by y', e.max(axis=(0,2,3))[:8], e.max(axis=(0,2,3))[-4:])
print('mean abs err', e.mean(), 'mean abs val', np.abs(nhwc(t64.grad)).mean())
ratio = draw/ (nhwc(t64.grad)+1e-30)
print('ratio pct', np.percentile(ratio, [1,25,50,75,99]))
