"""Stand-alone timing of the grouped BatchNorm kernels (favae_gbn_fwd / favae_gbn_bwd) against a plain device copy of the same
bytes.  Run on the GPU box: python tools/bench_gbn.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)
from favae_hip import lib as L  # noqa: E402

raw_ = L.raw()
_i, _ll, _f = ctypes.c_int, ctypes.c_longlong, ctypes.c_float


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, Rg, C in (("final_layer.0", 64 * 4096, 64), ("enc.0/dec.3", 64 * 1024, 64), ("enc.1/dec.2", 64 * 256, 128), ("enc.2/dec.1", 64 * 64, 256), ("enc.3/dec.0", 64 * 16, 512), ("enc.4", 64 * 4, 1024)):
    R = G * Rg
    raw = torch.randn(R, C, device="cuda")
    dact = torch.randn(R, C, device="cuda")
    act = torch.empty(R, C, device="cuda", dtype=torch.bfloat16)
    draw = torch.empty(R, C, device="cuda", dtype=torch.bfloat16)
    nb = 4
    while nb < 32 and nb * 16 * 64 < Rg:
        nb *= 2
    bins = torch.zeros(G, nb, 2, C, device="cuda", dtype=torch.float64)
    gbins = torch.zeros(G, nb, 2, C, device="cuda", dtype=torch.float64)
    gamma = torch.ones(C, device="cuda")
    beta = torch.zeros(C, device="cuda")
    mean = torch.zeros(G, C, device="cuda")
    invstd = torch.zeros(G, C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    dg, db, dbias = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    st = L.stream_ptr()
    ab_t = torch.zeros(G, 2, C, device="cuda")
    AB = L.ptr(ab_t) if os.environ.get("GBN_AB", "1") == "1" else None
    if C >= 64:
        L.check(raw_.favae_gbn_sums(L.ptr(raw), None, None, None, None, None, L.ptr(bins), _i(nb), _i(G), _ll(Rg), _i(C), _i(1), _i(L.BF16), st))
    else:
        x = raw.view(G, Rg, C).double()
        bins[:, 0, 0] = x.sum(1)
        bins[:, 0, 1] = (x * x).sum(1)

    def fwd():
        L.check(raw_.favae_gbn_fwd(L.ptr(raw), L.ptr(bins), _i(nb), L.ptr(gamma), L.ptr(beta), L.ptr(act), L.ptr(mean), L.ptr(invstd), AB, L.ptr(rm),
                                   L.ptr(rv), _i(G), _ll(Rg), _i(C), _f(1e-5), _f(0.1), _i(L.BF16), st))

    def bwd():
        L.check(raw_.favae_gbn_bwd(L.ptr(dact), L.ptr(act), L.ptr(raw), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), AB, L.ptr(gbins), _i(nb), L.ptr(draw),
                                   L.ptr(dg), L.ptr(db), (L.ptr(dbias) if os.environ.get("GBN_DBIAS", "1") == "1" else None), _i(G), _ll(Rg), _i(C), _i(L.BF16), st))

    def sums2():
        L.check(raw_.favae_gbn_sums(L.ptr(dact), L.ptr(act), L.ptr(raw), L.ptr(mean), L.ptr(invstd), AB, L.ptr(gbins), _i(nb), _i(G), _ll(Rg), _i(C), _i(2),
                                    _i(L.BF16), st))

    fwd()
    n = R * C
    tf, tb = timeit(fwd), timeit(bwd)
    tcopy = timeit(lambda: dact.copy_(raw))
    line = "%-14s G=%d R=%8d C=%3d: fwd %7.1f us %5.2f TB/s | bwd %7.1f us %5.2f TB/s | fp32 copy %7.1f us %5.2f TB/s" % (
        name, G, R, C, tf, 6 * n / tf / 1e6, tb, 12 * n / tb / 1e6, tcopy, 8 * n / tcopy / 1e6)
    if C >= 64:
        ts = timeit(sums2)
        line += " | sums2 %7.1f us %5.2f TB/s" % (ts, 10 * n / ts / 1e6)
    print(line, flush=True)
