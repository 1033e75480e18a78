"""Data-parallel VAE epoch on UNEVEN shards (N = 1001 over 2 ranks: 501 + 500 rows) against a single-process replay that
computes each rank's mini-batch gradient separately, averages them and applies Adam (developer tool + test driver).
Launch: python -m torch.distributed.run --nproc-per-node 2 tools/dp_uneven.py   (FAVAE_DIST_BACKEND=gloo on a one-GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "fa-vae_amd", "condition_a_VAE")):
    sys.path.insert(0, p)
os.environ["FAVAE_CONV_DTYPE"] = "f32"
import numpy as np, torch
import torch.distributed as td

backend = os.environ.get("FAVAE_DIST_BACKEND", "nccl")
rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ["LOCAL_RANK"] = "0" if backend != "nccl" else os.environ.get("LOCAL_RANK", "0")
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
td.init_process_group(backend)
from favae_hip import dist, imgvae
from favae_hip.engine import VaeEngine

N, z, B, lr = 1001, 8, 64, 1e-3
img = np.random.RandomState(1).rand(N, 3, 64, 64).astype(np.float32)
bounds = [dist.shard_bounds(N, ws, r) for r in range(ws)]
lo, hi = bounds[rank]


def fresh():
    torch.manual_seed(0)
    return imgvae.ImgVAE(dimx=z, channels=3, lr=lr, h=64, w=64, dataset="dp", precision="f32")


vae = fresh()
sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
torch.manual_seed(100 + rank)
vae.trainloop(img[lo:hi], epochs=1, favae=False)
got = {k: v.clone() for k, v in vae.state_dict().items()}
rec_dp, kl_dp = float(vae.reconstruc_during_training[-1]), float(vae.KL_QandP[-1])
td.barrier()
if rank == 0:
    # replay: one engine per virtual rank, gradients averaged by hand
    n_use = min(h - l for l, h in bounds)
    engs, datas, perms, epss = [], [], [], []
    for r, (l, h) in enumerate(bounds):
        e = VaeEngine(z, 3, batch_max=B, dtype="f32")
        e.load_state_dict(sd0)
        engs.append(e)
        datas.append(e.upload_images(img[l:h]))
        torch.manual_seed(100 + r)
        torch.empty((), dtype=torch.int64).random_()
        g = torch.Generator(); g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        perms.append(torch.randperm(h - l, generator=g)[:n_use].to(torch.int32).cuda())
        epss.append(torch.cat([torch.randn(min(B, n_use - s), z) for s in range(0, n_use, B)], 0).cuda())
    recs, kls = [], []
    for s in range(0, n_use, B):
        b = min(B, n_use - s)
        ms = [e.train_fwdbwd(d, p[s:s + b], b, ep[s:s + b], None, 1.0, 1.0, 0.01).cpu().numpy().copy()
              for e, d, p, ep in zip(engs, datas, perms, epss)]
        torch.cuda.synchronize()
        G = sum(e.G for e in engs) / len(engs)
        for e in engs:
            e.G.copy_(G)
            e.adam(lr)
        recs.append(np.mean([m[0] for m in ms])); kls.append(np.mean([m[1] for m in ms]))
    want = engs[0].state_dict()
    rec_ref, kl_ref = float(np.mean(recs)), float(np.mean(kls))
    print("shards %s, %d samples per rank and epoch, %d steps" % (bounds, n_use, len(recs)))
    print("epoch reconstruction term: dp %.6f replay %.6f rel %.3e" % (rec_dp, rec_ref, abs(rec_dp / rec_ref - 1)))
    print("epoch KL term: dp %.6f replay %.6f rel %.3e" % (kl_dp, kl_ref, abs(kl_dp / kl_ref - 1)))
    worst, frac = 0.0, 0.0
    for k in want:
        if "running" in k or "num_batches" in k or k.endswith(".0.bias"):
            continue
        d = (got[k].float().cpu() - want[k].float().cpu()).abs()
        worst = max(worst, float(d.max())); frac = max(frac, float((d > 1e-4).float().mean()))
    print("parameters after the epoch: max abs difference %.3e, largest share of entries off by > 1e-4: %.4f" % (worst, frac))
td.barrier()
dist.shutdown()
td.destroy_process_group()
