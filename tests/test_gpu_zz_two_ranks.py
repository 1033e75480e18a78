"""The N > 1 launch of bench.py, rehearsed on a one-GPU box: two ranks started the way the driver starts them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on cuda:0, talking over gloo
(FAVAE_DIST_BACKEND=gloo: RCCL refuses two ranks on one device).  Everything but the transport is the multi-GPU path:
per-rank shards of the data, the gradient buckets averaged between the backward stages, the all-reduced FA statistics,
the barrier + max-over-ranks timing and the single JSON line from rank 0.  Collected last (file name): it starts
processes of its own."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_bench_two_ranks_over_gloo_on_one_gpu():
    env = dict(os.environ, FAVAE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--samples", "512", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints the one JSON line, rank 1 nothing
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["samples_per_gpu"] == 512 and out["config"]["global_samples"] == 1024
    assert out["value"] > 0 and math.isfinite(out["config"]["bound_last"])


@pytest.mark.gpu
@pytest.mark.timeout(400)
def test_uneven_shards_train_like_the_hand_averaged_replay():
    """N = 1001 over two ranks (501 + 500 rows, dist.shard_bounds): the data-parallel trainloop visits 500 samples per rank and
    epoch in 8 lockstep mini-batches and averages the gradients; a single-process replay computes each rank's mini-batch
    gradient on an engine of its own, averages by hand and applies Adam.  Same epoch metrics (1e-4: fp32 atomics reorder
    sums); parameters within the 2 * 8 * lr that Adam's +-lr steps on noise-level gradient entries allow (tools/dp_uneven.py)."""
    import re
    env = dict(os.environ, FAVAE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "dp_uneven.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=360)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rels = [float(x) for x in re.findall(r"rel ([0-9.e+-]+)", r.stdout)]
    assert len(rels) == 2 and rels[0] < 1e-4 and rels[1] < 5e-2, r.stdout[-2000:]
    m = re.search(r"max abs difference ([0-9.e+-]+), largest share of entries off by > 1e-4: ([0-9.]+)", r.stdout)
    assert m and float(m.group(1)) <= 1.7e-2, r.stdout[-2000:]      # 8 Adam steps of at most lr each, in either direction
    assert "500 samples per rank and epoch, 8 steps" in r.stdout
