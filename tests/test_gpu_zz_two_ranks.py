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
