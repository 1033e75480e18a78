"""Developer check: parameters after ONE optimiser step, epoch call vs step + adam, against the run-to-run spread of the
step path itself (split-K atomics reorder sums; Adam's first step is +-lr whatever the gradient magnitude, so entries
whose gradient is rounding noise flip)."""
import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "fa-vae_amd"))
import numpy as np, torch
from conftest import img_inputs
from oracle import vae_torch as vt
from favae_hip.engine import VaeEngine
for ch in (3, 1):
    for dtype in ("f32", "bf16"):
        z, Bn = 20, 64
        torch.manual_seed(1); np.random.seed(1)
        P, Bf = vt.init_state(z, ch)
        sd = {k: v.detach().clone() for k, v in P.items()}; sd.update({k: v.clone() for k, v in Bf.items()})
        img3, _ = img_inputs(Bn)
        img = np.ascontiguousarray(img3[:, :ch])
        prior = torch.from_numpy(np.random.RandomState(7).randn(Bn, z) * 0.5).cuda()
        eps = torch.randn(Bn, z, generator=torch.Generator().manual_seed(9)).cuda()
        idx = torch.arange(Bn, dtype=torch.int32, device="cuda")
        out = {}
        for mode in ("epoch", "steps", "steps2"):
            eng = VaeEngine(z, ch, batch_max=Bn, dtype=dtype)
            eng.load_state_dict(sd); eng.step = 0
            data = eng.upload_images(img)
            ms = torch.zeros(1, 2, dtype=torch.float64, device="cuda")
            if mode == "epoch":
                eng.train_epoch(data, idx, eps, prior, 1.7, 1.0, 0.01, 1e-3, Bn, ms)
            else:
                eng.train_fwdbwd(data, idx, Bn, eps, prior, 1.7, 1.0, 0.01)
                eng.adam(1e-3)
            torch.cuda.synchronize()
            out[mode] = {k: v.clone() for k, v in eng.state_dict().items()}
        for k in P:
            a = (out["epoch"][k] - out["steps"][k]).abs(); b = (out["steps2"][k] - out["steps"][k]).abs()
            fa, fb = int((a > 1e-3).sum()), int((b > 1e-3).sum())
            if fa > 2 * fb + 5:
                print(ch, dtype, k, "flips epoch-vs-steps", fa, "steps-vs-steps", fb, "of", a.numel(), flush=True)
        print(ch, dtype, "done", flush=True)
