"""Grouped-batch mode: samples/s of the epoch entry point at G = 1, 2, 4, 8 BatchNorm groups per step (bf16 and fp32).
Run on the GPU box: python tools/bench_grouped.py [n [dtypes [groups]]]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)
from favae_hip.engine import VaeEngine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
z = 20
rng = np.random.RandomState(0)
img = torch.from_numpy(rng.randint(0, 256, size=(n, 64, 64, 3), dtype=np.uint8)).cuda()
dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ("bf16", "f32")
groups = [int(g) for g in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 2, 4, 8, 16)
for dtype in dtypes:
    for G in groups:
        batch = 64 * G
        eng = VaeEngine(z, 3, batch_max=batch, dtype=dtype)
        torch.manual_seed(0)
        eng.P.normal_(0, 0.02)
        sd = eng.state_dict()
        for k in sd:
            if k.endswith("running_var") or (k.endswith(".1.weight")):
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
        print("%s G=%2d batch %4d: %.3f ms/step  %.0f samples/s  (last loss %.4g)" % (dtype, G, batch, 1e3 * dt / ms.shape[0], n / dt, ms[-1].sum().item()), flush=True)
        del eng
