"""Rehearsal of the data-parallel trainloop on ONE GPU (developer tool): a one-rank RCCL group with FAVAE_FORCE_DP=1 takes
the multi-rank code path (staged backward, bucketed asynchronous all-reduce on the communication stream, replicated Adam),
so the stream ordering is exercised for real; prints ms/step next to the single-device epoch path."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fa-vae_amd"))
import numpy as np, torch
import torch.distributed as td
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=torch.device("cuda:0"))
from favae_hip import imgvae
rs = np.random.RandomState(1)
img = rs.rand(N, 3, 64, 64).astype(np.float32)
res = {}
from favae_hip import dist as fdist
for mode in ("epoch", "dp_native", "dp_torch"):
    os.environ["FAVAE_FORCE_DP"] = "0" if mode == "epoch" else "1"
    os.environ["FAVAE_NATIVE_COMM"] = "0" if mode == "dp_torch" else "1"
    fdist.shutdown()
    torch.manual_seed(0)
    vae = imgvae.ImgVAE(dimx=20, channels=3, lr=1e-3, dataset="rehearsal")
    vae.trainloop(img, epochs=1)
    torch.cuda.synchronize()
    t = time.time()
    vae.trainloop(img, epochs=2)
    torch.cuda.synchronize()
    dt = time.time() - t
    res[mode] = (dt, list(vae.elbo_training))
    if mode == "dp_native":
        print("native communicator in use:", fdist.native_comm(torch.device("cuda:0")) is not None, flush=True)
    print("%s: %.3f ms/step  elbo %s" % (mode, 1e3 * dt / (2 * ((N + 63) // 64)), vae.elbo_training), flush=True)
a = np.array(res["epoch"][1])
for mode in ("dp_native", "dp_torch"):
    b = np.array(res[mode][1])
    print("relative ELBO difference epoch vs %s path:" % mode, float(np.abs(a - b).max() / np.abs(a).max()))
fdist.shutdown()
td.destroy_process_group()
