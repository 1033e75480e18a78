import os, sys
sys.path.insert(0, '/root/repo/fa-vae_amd'); sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from test_gpu_vae_step import _engine, img_inputs, vt
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
z, Bn, N = 20, 64, 6 * 64 + 17
torch.manual_seed(0)
P, Bf = vt.init_state(z, 3)
sd = {k: v.detach().clone() for k, v in P.items()}
sd.update({k: v.clone() for k, v in Bf.items()})
img, _ = img_inputs(N)
perm = torch.randperm(N, generator=torch.Generator().manual_seed(5)).to(torch.int32).cuda()
eps = torch.randn(N, z, generator=torch.Generator().manual_seed(3)).cuda()
prior = torch.from_numpy(np.random.RandomState(7).randn(N, z) * 0.5).cuda()
out = {}
for mode in ("0", "0b", "1", "1b", "1e"):
    os.environ["FAVAE_ENQUEUE_THREAD"] = mode[0]
    os.environ["FAVAE_SYNC_VALUES"] = "0" if mode == "1e" else "1"
    eng = _engine(z, dtype)
    eng.load_state_dict(sd)
    data = eng.upload_images(img)
    ms = torch.zeros(2, 7, 2, dtype=torch.float64, device="cuda")
    for ep in range(2):
        eng.train_epoch(data, perm, eps, prior, 1.7, 1.0, 0.01, 1e-3, Bn, ms[ep])
    torch.cuda.synchronize()
    out[mode] = ms.cpu().numpy()
for a, b in (("0", "0b"), ("1", "1b"), ("0", "1"), ("0", "1e"), ("1", "1e")):
    m0, m1 = out[a], out[b]
    print(a, b, 'first-step rel', np.abs(m1[0, 0] / m0[0, 0] - 1).max(), 'rec max rel', np.abs(m1[:, :, 0] / m0[:, :, 0] - 1).max(),
          'KL median |log ratio|', np.median(np.abs(np.log(m1[:, :, 1] / m0[:, :, 1]))))
print(out["0"][:, :, 1])
print(out["1"][:, :, 1])
