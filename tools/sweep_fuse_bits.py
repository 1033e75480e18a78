"""Every FAVAE_FUSE_MASK switch of the executor toggled on its own: a 3-step epoch (bf16 and fp32) must give finite loss terms whose
first step agrees with the default path.  Run on the GPU box: python tools/sweep_fuse_bits.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)

BITS = [1, 2, 4, 8, 16, 32, 64, 128, 0x100, 0x200, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000, 0x20000, 0x40000, 0x80000, 0x100000,
        0x400000, 0x2000000, 0x4000000, 0x8000000, 0x10000000, 0x20000000, 0x40000000, 0x80000000]


def run(mask, dtype, n=192, z=20):
    os.environ["FAVAE_FUSE_MASK"] = hex(mask)
    from favae_hip.engine import VaeEngine
    eng = VaeEngine(z, 3, batch_max=64, dtype=dtype)
    torch.manual_seed(0)
    eng.P.normal_(0, 0.02)
    sd = eng.state_dict()
    for k in sd:
        if k.endswith("running_var") or k.endswith(".1.weight"):
            sd[k].fill_(1.0)
    eng.load_state_dict(sd)
    img = torch.from_numpy(np.random.RandomState(0).randint(0, 256, size=(n, 64, 64, 3), dtype=np.uint8)).cuda()
    perm = torch.arange(n, dtype=torch.int32).cuda()
    eps = torch.randn(n, z, generator=torch.Generator().manual_seed(1)).cuda()
    prior = torch.zeros(n, z, dtype=torch.float64).cuda()
    ms = torch.zeros(n // 64, 2, dtype=torch.float64, device="cuda")
    eng.train_epoch(img, perm, eps, prior, 1.0, 1.0, 0.01, 1e-4, 64, ms)
    torch.cuda.synchronize()
    xm = torch.zeros(n, z, dtype=torch.float64, device="cuda")
    xc = torch.zeros(n, z, dtype=torch.float64, device="cuda")
    eng.encode_all(img, xm, xc)
    acc = torch.zeros(3, dtype=torch.float64, device="cuda")
    eng.reconstruct_stats(torch.randn(n, z, generator=torch.Generator().manual_seed(2)).cuda(), img, acc)
    torch.cuda.synchronize()
    return ms.cpu().numpy(), float(xm.abs().sum()), float(acc[0])


def sweep(verbose=True):
    """-> list of (dtype, bit, note) of the switches that do not reproduce the default path"""
    old = os.environ.get("FAVAE_FUSE_MASK")
    bad = []
    try:
        for dtype in ("bf16", "f32"):
            ref, rx, ra = run(0x17f, dtype)
            for b in BITS:
                try:
                    m, x, a = run(0x17f ^ b, dtype)
                    ok = (np.isfinite(m).all() and np.allclose(m[0], ref[0], rtol=2e-3) and np.allclose(m, ref, rtol=0.1) and abs(x / rx - 1) < 0.10
                          and abs(a / ra - 1) < 0.05)   # (|X|: eval-mode latent means after three steps, running statistics far from settled: 5.3 % seen between two healthy paths)
                    note = "" if ok else "  first step %s vs %s; last %s vs %s; |X| %.4g vs %.4g; sse %.4g vs %.4g" % (m[0], ref[0], m[-1], ref[-1], x, rx, a, ra)
                except Exception as e:   # noqa: BLE001
                    ok, note = False, "  raised: %s" % str(e)[:200]
                if not ok:
                    bad.append((dtype, b, note))
                if verbose:
                    print("%s  mask %#x (bit %#x %s): %s%s" % (dtype, 0x17f ^ b, b, "off" if 0x17f & b else "on", "ok" if ok else "MISMATCH", note), flush=True)
    finally:
        if old is None:
            os.environ.pop("FAVAE_FUSE_MASK", None)
        else:
            os.environ["FAVAE_FUSE_MASK"] = old
    return bad


if __name__ == "__main__":
    bad = sweep()
    print("switches that do not reproduce the default: %d" % len(bad))
    sys.exit(1 if bad else 0)
