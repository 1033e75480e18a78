import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/fa-vae_amd')
import numpy as np, torch
from conftest import img_inputs
from test_gpu_sshiba import load_variant
from oracle import fa_numpy as fo, vae_torch as vt
fav = load_variant("condition_a_VAE")
N=32; img, lab = img_inputs(N)
ov=[int(x) for x in sys.argv[1].split(',')] if len(sys.argv)>1 else [2,1]
np.random.seed(0); torch.manual_seed(0)
ref = fo.SSHIBAOracle(4,1,variant="A", vae_factory=lambda **kw: vt.OracleImgVAE(**kw)); ref.epochs_override=ov
ref.fit(fo.struct_data(img,"img",latent_dim=8,lr=1e-3,dataset="g"), fo.struct_data(lab,"mult"), max_iter=len(ov), pruning_crit=1e-3)
np.random.seed(0); torch.manual_seed(0)
model = fav.SSHIBA(4,1); model.epochs_override=ov
model.fit(model.struct_data(img,"img",latent_dim=8,lr=1e-3,dataset="g"), model.struct_data(lab,"mult"), max_iter=len(ov), pruning_crit=1e-3, verbose=0, store=False, pretrained=0)
def rel(a,b):
    a=np.asarray(a,dtype=np.float64); b=np.asarray(b,dtype=np.float64); return np.abs(a-b).max()/(np.abs(b).max()+1e-300)
print('L', model.L, ref.L)
for i in range(len(ov)): print('elbo', i, rel(model.total_elbo[i], ref.total_elbo[i]), model.total_elbo[i], ref.total_elbo[i])
print('Xmean', rel(model.X[0]['mean'], ref.X[0]['mean']), 'Xcov', rel(model.X[0]['cov'], ref.X[0]['cov']))
print('Z', rel(model.q_dist.Z['mean'], ref.q_dist.Z['mean']))
for m in range(2):
    print('W',m, rel(model.q_dist.W[m]['mean'], ref.q_dist.W[m]['mean']), 'tau', model.q_dist.tau[m], ref.q_dist.tau[m], 'b', rel(model.q_dist.b[m]['mean'], ref.q_dist.b[m]['mean']))
print(model.wandb_metrics); print(ref.metrics)
