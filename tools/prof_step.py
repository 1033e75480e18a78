"""Time / profile the HIP conv-VAE training epoch on synthetic data (developer tool)."""
import sys, time
sys.path.insert(0, '/root/repo/fa-vae_amd')
import numpy as np, torch
from favae_hip.engine import VaeEngine
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 1      # BatchNorm groups of 64 per step (grouped-batch mode)
B = 64 * groups
z = 20
eng = VaeEngine(z, 3, dtype=dtype) if groups == 1 else VaeEngine(z, 3, batch_max=B, dtype=dtype)
torch.manual_seed(0)
sys.path.insert(0, '/root/repo')
from oracle import vae_torch as vt
P, Bf = vt.init_state(z, 3)
sd = {k: v.detach() for k, v in P.items()}; sd.update(Bf)
eng.load_state_dict(sd)
rs = np.random.RandomState(1)
img = rs.rand(N, 3, 64, 64).astype(np.float32)
data = eng.upload_images(img)
perm = torch.randperm(N).to(torch.int32).cuda()
eps = torch.randn(N, z).cuda()
prior = torch.randn(N, z, dtype=torch.float64).cuda()
nsteps = (N + B - 1) // B
metrics = torch.zeros(nsteps, 2, dtype=torch.float64, device='cuda')
import os
stream = torch.cuda.Stream(priority=int(os.environ.get('FAVAE_MAIN_PRIO', '0'))) if os.environ.get('FAVAE_OWN_STREAM', '1') == '1' else torch.cuda.current_stream()
torch.cuda.synchronize()
torch.cuda.set_stream(stream)
eng.train_epoch(data, perm, eps, prior, 1.5, 1.0, 0.01, 1e-3, B, metrics)
torch.cuda.synchronize()
for r in range(reps):
    t = time.time()
    eng.train_epoch(data, perm, eps, prior, 1.5, 1.0, 0.01, 1e-3, B, metrics)
    t_enq = time.time() - t
    torch.cuda.synchronize()
    dt = time.time() - t
    print('host enqueue %.3f ms/step' % (1e3 * t_enq / nsteps))
    print('%s epoch N=%d: %.3f s  %.3f ms/step  %.0f samples/s  loss %.1f' % (dtype, N, dt, 1e3 * dt / nsteps, N / dt, metrics[-1, 0].item()))
