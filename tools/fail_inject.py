"""Failure injection for the threaded epochs (developer tool / test helper, own process): FAVAE_DEBUG_FAIL_JOB=n makes the n-th
job of the enqueue worker fail without issuing its calls, so that stream-word waits of that step are left without their
writes.  The epoch call must RETURN an error (not hang in hipDeviceSynchronize behind a wait that never ends), and the
executor must be usable afterwards.  argv[1] = 'single' (favae_vae_train_epoch) or 'dp' (one-rank RCCL group,
favae_vae_train_epoch_dp: the communication stream holds waits on the bucket fences)."""
import os
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "single"
os.environ["FAVAE_DEBUG_FAIL_JOB"] = sys.argv[2] if len(sys.argv) > 2 else "23"
os.environ["FAVAE_SYNC_VALUES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fa-vae_amd"))
import numpy as np
import torch

torch.cuda.set_device(0)
if mode == "dp":
    import torch.distributed as td
    td.init_process_group("nccl", device_id=torch.device("cuda:0"))
    os.environ["FAVAE_FORCE_DP"] = "1"
from favae_hip import lib as L
from favae_hip.engine import VaeEngine

z, N, B = 20, 8 * 64, 64
torch.manual_seed(0)
eng = VaeEngine(z, 3, batch_max=B, dtype="bf16")
eng.P.normal_(0, 0.02)
eng.refresh_weights()
img = np.random.RandomState(1).rand(N, 3, 64, 64).astype(np.float32)
data = eng.upload_images(img)
perm = torch.randperm(N).to(torch.int32).cuda()
eps = torch.randn(N, z).cuda()
ms = torch.zeros(N // B, 2, dtype=torch.float64, device="cuda")
comm = None
if mode == "dp":
    from favae_hip import dist
    comm = dist.native_comm(torch.device("cuda:0"))
    assert comm is not None, "no native communicator"
try:
    if comm is None:
        eng.train_epoch(data, perm, eps, None, 1.0, 1.0, 0.01, 1e-3, B, ms)
    else:
        eng.train_epoch_dp(comm, data, perm, eps, None, 1.0, 1.0, 0.01, 1e-3, B, ms)
    print("epoch returned without an error", flush=True)
    sys.exit(3)
except L.FavaeHipError as e:
    print("epoch returned an error:", str(e)[:120], flush=True)
    assert "injected failure" in str(e)
torch.cuda.synchronize()
print("device synchronised after the failure", flush=True)
# the executor still works (this epoch runs without the worker, so nothing is injected)
os.environ["FAVAE_ENQUEUE_THREAD"] = "0"
eng.P.normal_(0, 0.02)
eng.M.zero_(); eng.V.zero_(); eng.G.zero_()
eng.refresh_weights()
eng.step = 0
ms.zero_()
eng.train_epoch(data, perm, eps, None, 1.0, 1.0, 0.01, 1e-3, B, ms)
torch.cuda.synchronize()
m = ms.cpu().numpy()
assert np.isfinite(m).all() and (m[:, 0] > 0).all(), m
print("recovered: a later epoch runs, reconstruction term of its last step %.1f" % m[-1, 0], flush=True)
if mode == "dp":
    from favae_hip import dist
    dist.shutdown()
    td.destroy_process_group()
