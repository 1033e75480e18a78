"""Wall-clock of the phases of one FA-VAE iteration (developer tool): python tools/phase_times.py [N]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "fa-vae_amd", "condition_a_VAE")):
    sys.path.insert(0, p)
os.environ.setdefault("FAVAE_CONV_DTYPE", "bf16")
import numpy as np, torch
if os.environ.get("FAVAE_FORCE_DP") == "1":      # the multi-rank code path in a one-rank RCCL group
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29519"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", device_id=torch.device("cuda:0"))
import favae
from favae_hip import imgvae, sshiba
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(1)
img = rng.random((N, 3, 64, 64), dtype=np.float32)
rs = np.random.RandomState(1)
lab = (rs.randn(N, 5) @ rs.randn(5, 40) > 0).astype(int)
np.random.seed(0); torch.manual_seed(0)
model = favae.SSHIBA(50, 1)
model.fit(model.struct_data(img, "img", latent_dim=20, lr=1e-3, dataset="s"), model.struct_data(lab, "mult"), max_iter=0,
          pruning_crit=1e-3, verbose=0, store=False, pretrained=1)
T = {}
def timed(cls, name):
    f = getattr(cls, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.time()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.time() - t
        return r
    setattr(cls, name, g)
for n in ("update_w", "update_Z", "update_b", "update_tau", "update_alpha", "update_x", "update_bound", "pruning", "depruning",
          "_refresh_xstats", "_zero_input_r2"):
    timed(sshiba.SSHIBA, n)
for n in ("trainloop", "update_x", "reconstruction_r2", "restore_weights", "_prior_rows"):
    timed(imgvae.ImgVAE, n)
from favae_hip import engine as _engine
for n in ("train_epoch", "train_epoch_dp", "snapshot"):
    timed(_engine.VaeEngine, n)
def it():
    model.keep_training[0] = True
    model.update(iter_count=1); model.depruning(1e-15); model.L.append(model.update_bound()); model.pruning(1e-3)
it(); T.clear()
torch.cuda.synchronize(); t0 = time.time(); it(); torch.cuda.synchronize(); tot = time.time() - t0
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("%-22s %8.3f s" % (k, v))
print("iteration %.3f s (N=%d)" % (tot, N))
