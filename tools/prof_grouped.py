"""Two epochs of the grouped-batch mode at G groups per step (default 8, bf16) for rocprofv3 --kernel-trace --stats.
Run on the GPU box: rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/prof_grouped.py [G] [n] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)
from favae_hip.engine import VaeEngine  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
z, batch = 20, 64 * G
img = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(n, 64, 64, 3), dtype=np.uint8)).cuda()
eng = VaeEngine(z, 3, batch_max=batch, dtype=dtype)
torch.manual_seed(0)
eng.P.normal_(0, 0.02)
sd = eng.state_dict()
for k in sd:
    if k.endswith("running_var") or k.endswith(".1.weight"):
        sd[k].fill_(1.0)
eng.load_state_dict(sd)
perm = torch.randperm(n).to(torch.int32).cuda()
eps = torch.randn(n, z).cuda()
prior = torch.zeros(n, z, dtype=torch.float64).cuda()
ms = torch.zeros(eng.epoch_steps(n, batch), 2, dtype=torch.float64, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_epoch(img, perm, eps, prior, 1.0, 1.0, 0.01, 1e-5, batch, ms)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("G=%d %s: %.3f ms/step %.0f samples/s" % (G, dtype, 1e3 * dt / ms.shape[0], n / dt), flush=True)
