"""Round-3 diagnostic: measured errors behind the tolerances of tests/test_gpu_vae_step.py and the bf16 coordinator test.
Prints, for fp32 and bf16: training-mode mu / std error of one step, eval-mode mu / var error, per-tensor gradient p99.9 and cosine.
Run on the GPU box: python tools/dbg_parity_r3.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from conftest import img_inputs  # noqa: E402
from oracle import vae_torch as vt  # noqa: E402
from favae_hip.engine import VaeEngine  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def p999(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.percentile(np.abs(a - b), 99.9) / (np.abs(b).max() + 1e-30)


def cos(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


NOISE_ONLY = {"enc.%d.0.bias" % i for i in range(5)} | {"dec.%d.0.bias" % i for i in range(4)} | {"final_layer.0.bias"}

for z, Bn in ((20, 16), (20, 64), (100, 64)):
    for dtype in ("f32", "bf16"):
        torch.manual_seed(0)
        np.random.seed(0)
        P, Bf = vt.init_state(z, 3)
        names = list(P.keys())
        opt = torch.optim.Adam([P[k] for k in names], 1e-3)
        eng = VaeEngine(z, 3, batch_max=64, dtype=dtype)
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
            opt.zero_grad()
            loss, rec_err, kl, mu, std, zz, rec = vt.loss_terms(P, Bf, images, prior, tau, 0.01, 1, eps=eps)
            loss.backward()
            m = eng.train_fwdbwd(data, idx, Bn, eps.cuda(), prior.cuda(), float(tau.item()), 1.0, 0.01).cpu().numpy()
            fc = eng.debug(0, 5, (Bn, 2 * z), torch.float32).cpu().numpy()
            g = eng.unpack(eng.G)
            worst_p, worst_c = ("", 0.0), ("", 1.0)
            for k in names:
                if k in NOISE_ONLY:
                    continue
                pp, cc = p999(g[k].numpy(), P[k].grad.numpy()), cos(g[k].numpy(), P[k].grad.numpy())
                if pp > worst_p[1]:
                    worst_p = (k, pp)
                if cc < worst_c[1]:
                    worst_c = (k, cc)
            logvar = 2 * np.log(std.detach().numpy())
            print("z=%d B=%d %s step %d: rec %.2e kl %.2e | train mu %.2e (rms-rel %.2e) logvar-raw %.2e | grad p99.9 worst %s %.2e | cos worst %s %.5f"
                  % (z, Bn, dtype, step, abs(m[0] / rec_err.item() - 1), abs(m[1] / kl.item() - 1), rel(fc[:, :z], mu.detach().numpy()),
                     np.sqrt(((fc[:, :z] - mu.detach().numpy()) ** 2).mean()) / np.sqrt((mu.detach().numpy() ** 2).mean()),
                     rel(fc[:, z:] + 1e-6, logvar), worst_p[0], worst_p[1], worst_c[0], worst_c[1]), flush=True)
            opt.step()
        # eval mode on the trained state (running statistics after 3 updates)
        sd = {k: v.detach().clone() for k, v in P.items()}
        sd.update({k: v.clone() for k, v in Bf.items()})
        eng.load_state_dict(sd)
        xm = torch.zeros(Bn, z, dtype=torch.float64, device="cuda")
        xc = torch.zeros(Bn, z, dtype=torch.float64, device="cuda")
        eng.encode(data, None, Bn, xm, xc)
        with torch.no_grad():
            mu, std = vt.encoder(images, P, Bf, False)
        print("z=%d B=%d %s eval: mu %.2e var %.2e" % (z, Bn, dtype, rel(xm.cpu().numpy(), mu.numpy()), rel(xc.cpu().numpy(), std.pow(2).numpy())), flush=True)
        del eng

# eval mode with non-trivial running statistics (the state of test_eval_encoder_and_decoder_match_oracle)
for dtype in ("f32", "bf16"):
    z, Bn = 8, 24
    torch.manual_seed(3)
    P, Bf = vt.init_state(z, 3)
    for k in Bf:
        if k.endswith("running_var"):
            Bf[k].uniform_(0.5, 1.5)
        elif k.endswith("running_mean"):
            Bf[k].normal_(0, 0.1)
    eng = VaeEngine(z, 3, batch_max=64, dtype=dtype)
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.clone() for k, v in Bf.items()})
    eng.load_state_dict(sd)
    img, _ = img_inputs(Bn)
    data = eng.upload_images(img)
    xm = torch.zeros(Bn, z, dtype=torch.float64, device="cuda")
    xc = torch.zeros(Bn, z, dtype=torch.float64, device="cuda")
    eng.encode(data, None, Bn, xm, xc)
    with torch.no_grad():
        mu, std = vt.encoder(torch.tensor(img), P, Bf, False)
        zin = torch.randn(Bn, z, generator=torch.Generator().manual_seed(5))
        rec = vt.decoder(zin, P, Bf, False)
    out = torch.zeros(Bn, 64, 64, 3, device="cuda")
    eng.decode(zin.cuda(), Bn, out)
    print("eval-test state %s: mu %.2e var %.2e dec %.2e" % (dtype, rel(xm.cpu().numpy(), mu.numpy()), rel(xc.cpu().numpy(), std.pow(2).numpy()),
                                                            rel(out.cpu().permute(0, 3, 1, 2).numpy(), rec.numpy())), flush=True)
