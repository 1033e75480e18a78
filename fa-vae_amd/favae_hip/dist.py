"""Multi-GPU glue: one process per GPU, torch.distributed over RCCL (backend "nccl").  The path shards along N
(SURVEY.md 8e): per-sample state stays on its rank; only reduced statistics, VAE gradients, epoch metrics and BN running
statistics are exchanged.  All helpers are no-ops in a single process."""
import torch
import torch.distributed as td


import os


def active():
    # FAVAE_FORCE_DP=1: take the multi-rank code path in a one-rank group as well (rehearsal of the RCCL stream logic on
    # one GPU, tools/dp_rehearsal.py)
    if not (td.is_available() and td.is_initialized()):
        return False
    return td.get_world_size() > 1 or os.environ.get("FAVAE_FORCE_DP") == "1"


def world_size():
    return td.get_world_size() if active() else 1


def rank():
    return td.get_rank() if active() else 0


def _staged(t, fn):
    """gloo (CPU tests / one-GPU rehearsal of the N>1 path) works on host memory: stage device tensors through it."""
    if t.is_cuda and td.get_backend() != "nccl":
        h = t.cpu()
        fn(h)
        t.copy_(h)
    else:
        fn(t)
    return t


def _native_reduce(t, average):
    """All-reduce of a contiguous fp64 / fp32 device tensor through the library's own communicator (favae_allreduce_f64 / _f32), on
    the current stream: one communicator serves the gradient buckets of the VAE epochs and the FA statistics.  False when there is
    no such communicator (gloo rehearsals, FAVAE_NATIVE_COMM=0) or the tensor is not eligible."""
    if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float64, torch.float32) and t.numel() > 0):
        return False
    comm = native_comm(t.device)
    if comm is None:
        return False
    import ctypes
    from . import lib as L
    f = L.raw().favae_allreduce_f64 if t.dtype == torch.float64 else L.raw().favae_allreduce_f32
    L.check(f(comm, L.ptr(t), ctypes.c_longlong(t.numel()), ctypes.c_int(1 if average else 0), L.stream_ptr()))
    return True


def sum_(t):
    """In-place sum over ranks (RCCL all-reduce on the device buffer)."""
    if active() and not _native_reduce(t, False):
        _staged(t, lambda x: td.all_reduce(x, op=td.ReduceOp.SUM))
    return t


def sum_many_(ts):
    """In-place sum over ranks of several statistics in ONE all-reduce: the tensors of a phase are packed into one buffer,
    reduced and scattered back (small K x K / d x K blocks: one collective's latency instead of one per statistic)."""
    if active() and len(ts) > 0:
        if len(ts) == 1:
            return [sum_(ts[0])]
        flat = torch.cat([t.reshape(-1) for t in ts])
        sum_(flat)
        o = 0
        for t in ts:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
    return ts


def mean_(t):
    """In-place mean over ranks.  RCCL has a native AVG; gloo does not."""
    if active() and not _native_reduce(t, True):
        if td.get_backend() == "nccl":
            td.all_reduce(t, op=td.ReduceOp.AVG)
        else:
            _staged(t, lambda x: (td.all_reduce(x, op=td.ReduceOp.SUM), x.div_(td.get_world_size())))
    return t


_native = {"comm": None, "tried": False}


def native_comm(device):
    """The library's own RCCL communicator (favae_comm_*), bootstrapped through torch.distributed: rank 0 draws the RCCL
    unique id, a broadcast hands it to the others, every rank joins.  None when the backend is not RCCL or the bootstrap
    fails (the callers then take the torch.distributed path).  FAVAE_NATIVE_COMM=0 disables it."""
    if _native["tried"]:
        return _native["comm"]
    _native["tried"] = True
    if not active() or td.get_backend() != "nccl" or os.environ.get("FAVAE_NATIVE_COMM") == "0":
        return None
    import ctypes
    from . import lib as L
    raw = L.raw()
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
    ident = (ctypes.c_ubyte * 128)()
    ok = torch.ones(1, dtype=torch.int32, device=device)
    if td.get_rank() == 0 and raw.favae_comm_unique_id(path, ident) != 0:
        ok.zero_()
    t = torch.tensor(list(ident), dtype=torch.uint8, device=device)
    td.broadcast(t, 0)
    td.all_reduce(ok, op=td.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None
    ident = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
    comm = ctypes.c_void_p()
    rc = raw.favae_comm_init(ctypes.byref(comm), path, ident, ctypes.c_int(td.get_rank()), ctypes.c_int(td.get_world_size()))
    ok.fill_(1 if rc == 0 else 0)
    td.all_reduce(ok, op=td.ReduceOp.MIN)          # all ranks or none
    if int(ok.item()) == 0:
        if rc == 0:
            raw.favae_comm_destroy(comm)
        return None
    _native["comm"] = comm
    return comm


def shutdown():
    if _native["comm"] is not None:
        from . import lib as L
        L.raw().favae_comm_destroy(_native["comm"])
    _native["comm"], _native["tried"] = None, False


def mean_async(t, engine, comm_stream):
    """Mean over ranks of a gradient bucket whose producers (main + weight-gradient stream of `engine`) may still be running:
    RCCL: the all-reduce is enqueued behind them on `comm_stream` and returns a work handle (wait() orders the current
    stream behind it); other backends: blocking, staged through the host, returns None."""
    if not active():
        return None
    if td.get_backend() == "nccl":
        engine.fence(comm_stream)
        with torch.cuda.stream(comm_stream):
            return td.all_reduce(t, op=td.ReduceOp.AVG, async_op=True)
    engine.fence(torch.cuda.current_stream())
    _staged(t, lambda x: (td.all_reduce(x, op=td.ReduceOp.SUM), x.div_(td.get_world_size())))
    return None


def broadcast_(t, src=0):
    if active():
        _staged(t, lambda x: td.broadcast(x, src))
    return t


def sum_scalar(x, device=None):
    """Sum of a Python int/float over ranks."""
    if not active():
        return x
    t = torch.tensor([float(x)], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return t.item()


def masked_col_stats(A, keep, device=None):
    """Per-column count, mean and population standard deviation of the entries of A where `keep`, over the rows of ALL
    ranks: what np.nanmean / np.nanstd give the reference on the whole data set (cA/favae.py:251, 278-286).  Host arrays
    in and out; two passes (the mean first), fp64.  Columns without a kept entry come back as NaN."""
    import numpy as np

    def total(x):
        if not active():
            return x
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        if td.get_backend() == "nccl":
            t = t.to(device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return t.cpu().numpy()

    cs = total(np.stack([keep.sum(0).astype(np.float64), np.where(keep, A, 0.0).sum(0)]))
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = cs[1] / cs[0]
        dev2 = total(np.where(keep, (A - mean) ** 2, 0.0).sum(0))
        std = np.sqrt(dev2 / cs[0])
    return cs[0], mean, std


def min_scalar(x, device=None):
    """Minimum of a Python int/float over ranks."""
    if not active():
        return x
    t = torch.tensor([float(x)], dtype=torch.float64, device=device if td.get_backend() == "nccl" else "cpu")
    td.all_reduce(t, op=td.ReduceOp.MIN)
    return t.item()


def max_scalar(x, device=None):
    return -min_scalar(-x, device)


def all_equal(x, device=None):
    if not active():
        return True
    return sum_scalar(x, device) == x * world_size()


def shard_bounds(n, ws=None, rk=None):
    """Contiguous equal-as-possible shard [lo, hi) of n rows for this rank (np.array_split convention)."""
    ws = world_size() if ws is None else ws
    rk = rank() if rk is None else rk
    base, rem = divmod(n, ws)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)
