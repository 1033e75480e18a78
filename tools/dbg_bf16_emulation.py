"""Per-tensor agreement of the bf16 step with (a) the fp32 oracle and (b) the oracle with the executor's rounding points
(tests/bf16_emulation.py), at a point of training past the initialisation.  Run on the GPU box: python tools/dbg_bf16_emulation.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from conftest import img_inputs  # noqa: E402
from oracle import vae_torch as vt  # noqa: E402
from bf16_emulation import loss_terms_bf16  # noqa: E402
from favae_hip.engine import VaeEngine  # noqa: E402
from test_gpu_vae_step import NOISE_ONLY, _cos, _rel_p99  # noqa: E402

z, Bn = 20, 64
torch.manual_seed(0)
np.random.seed(0)
P, Bf = vt.init_state(z, 3)
names = list(P.keys())
opt = torch.optim.Adam([P[k] for k in names], 1e-3)
eng = VaeEngine(z, 3, batch_max=64, dtype="bf16")
img, _ = img_inputs(Bn)
prior = torch.from_numpy(np.random.RandomState(7).randn(Bn, z) * 0.5)
tau = torch.Tensor(np.asarray(np.array([1.7])))
images = torch.tensor(img)
data = eng.upload_images(img)
idx = torch.arange(Bn, dtype=torch.int32, device="cuda")
for step in range(3):
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.clone() for k, v in Bf.items()})
    eng.load_state_dict(sd)
    eps = torch.randn(Bn, z, generator=torch.Generator().manual_seed(100 + step))
    eng.train_fwdbwd(data, idx, Bn, eps.cuda(), prior.cuda(), float(tau.item()), 1.0, 0.01)
    g = {k: v.numpy().copy() for k, v in eng.unpack(eng.G).items()}
    res = {}
    for tag, fn in (("emu", loss_terms_bf16), ("f32", lambda *a: vt.loss_terms(*a[:-1], eps=a[-1]))):
        Bc = {k: v.clone() for k, v in Bf.items()}
        opt.zero_grad()
        out = fn(P, Bc, images, prior, tau, 0.01, 1, eps)
        out[0].backward()
        res[tag] = {k: P[k].grad.numpy().copy() for k in names}
    if step >= 1:
        print("step %d" % step)
        for k in names:
            if k in NOISE_ONLY:
                continue
            print("  %-24s emu: p99.9 %.2e cos %.5f | f32: p99.9 %.2e cos %.5f | emu vs f32 cos %.5f" % (
                k, _rel_p99(g[k], res["emu"][k]), _cos(g[k], res["emu"][k]), _rel_p99(g[k], res["f32"][k]), _cos(g[k], res["f32"][k]),
                _cos(res["emu"][k], res["f32"][k])), flush=True)
    # advance on the fp32 oracle's gradient
    opt.zero_grad()
    loss = vt.loss_terms(P, Bf, images, prior, tau, 0.01, 1, eps=eps)[0]
    loss.backward()
    opt.step()
