"""The two whole-data-set eval passes of an FA-VAE iteration (update_x = encoder, reconstruction + R^2 sums = decoder) over n
images, for timing and for rocprofv3 --kernel-trace.  Run on the GPU box: python tools/prof_eval.py [n] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)
from favae_hip.engine import VaeEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
z = 20
img = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(n, 64, 64, 3), dtype=np.uint8)).cuda()
eng = VaeEngine(z, 3, batch_max=64, dtype=dtype)
torch.manual_seed(0)
eng.P.normal_(0, 0.02)
sd = eng.state_dict()
for k in sd:
    if k.endswith("running_var") or k.endswith(".1.weight"):
        sd[k].fill_(1.0)
eng.load_state_dict(sd)
xm = torch.zeros(n, z, dtype=torch.float64, device="cuda")
xc = torch.zeros(n, z, dtype=torch.float64, device="cuda")
zin = torch.randn(n, z, device="cuda")
acc = torch.zeros(3, dtype=torch.float64, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.encode_all(img, xm, xc)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng.reconstruct_stats(zin, img, acc)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("update_x %.1f ms (%.0f k samples/s)   reconstruction %.1f ms (%.0f k samples/s)   eval batch %d"
          % (1e3 * (t1 - t0), n / (t1 - t0) / 1e3, 1e3 * (t2 - t1), n / (t2 - t1) / 1e3, eng.eval_batch), flush=True)
