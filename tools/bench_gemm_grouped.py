"""Stand-alone timing of the convolution GEMMs of one grouped step (B = 64 G images): forward and input-gradient launches of every
layer, with and without the BatchNorm sums in the epilogue.  Run on the GPU box: python tools/bench_gemm_grouped.py [G]"""
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
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 64 * G


def timeit(f, n=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


# (name, kind, Cin, Cout, Hin, Hout): kind c = conv s2, t = convT s2
LAYERS = [("enc.1", "c", 64, 128, 32, 16), ("enc.2", "c", 128, 256, 16, 8), ("enc.3", "c", 256, 512, 8, 4), ("enc.4", "c", 512, 1024, 4, 2),
          ("dec.0", "t", 1024, 512, 2, 4), ("dec.1", "t", 512, 256, 4, 8), ("dec.2", "t", 256, 128, 8, 16), ("dec.3", "t", 128, 64, 16, 32),
          ("final.0", "t", 64, 64, 32, 64)]
st = L.stream_ptr()
tot = {"fwd": 0.0, "fwd_red": 0.0, "dg": 0.0, "dg_red": 0.0}
for name, kind, Ci, Co, Hi, Ho in LAYERS:
    x = torch.randn(B, Hi, Hi, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    wt = (torch.randn(Ci, 9, Co, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.zeros(Co, device="cuda")
    out = torch.zeros(B, Ho, Ho, Co, device="cuda")
    dy = torch.randn(B, Ho, Ho, Co, device="cuda").to(torch.bfloat16)
    dx = torch.zeros(B, Hi, Hi, Ci, device="cuda")
    actin = torch.randn(B, Hi, Hi, Ci, device="cuda").to(torch.bfloat16)
    rawin = torch.randn(B, Hi, Hi, Ci, device="cuda")
    mean = torch.zeros(G, max(Ci, Co), device="cuda")
    invstd = torch.ones(G, max(Ci, Co), device="cuda")
    bins = torch.zeros(G * 32 * 2 * max(Ci, Co), device="cuda", dtype=torch.float64)
    fused = ctypes.c_int(0)
    sc_f = 1 if kind == "t" else 0
    sc_b = 0 if kind == "t" else 1

    def nb(rows):
        n = 4
        while n < 32 and n * 16 * 64 < rows:
            n *= 2
        return n

    def fwd():
        L.check(raw_.favae_conv_igemm(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(out), _i(B), _i(Hi), _i(Hi), _i(Ci), _i(Ho), _i(Ho), _i(Co), _i(3), _i(2),
                                      _i(1), _i(sc_f), _i(0), _i(L.BF16), st))

    def fwd_red():
        L.check(raw_.favae_conv_igemm_red_groups(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(out), _i(B), _i(Hi), _i(Hi), _i(Ci), _i(Ho), _i(Ho), _i(Co), _i(3),
                                                 _i(2), _i(1), _i(sc_f), _i(0), _i(L.BF16), L.ptr(bins), _i(nb(64 * Ho * Ho)), _i(1), None, None, None, None, None,
                                                 _i(64), ctypes.byref(fused), st))

    def dg():
        L.check(raw_.favae_conv_igemm(L.ptr(dy), L.ptr(wt), None, L.ptr(dx), _i(B), _i(Ho), _i(Ho), _i(Co), _i(Hi), _i(Hi), _i(Ci), _i(3), _i(2), _i(1),
                                      _i(sc_b), _i(0), _i(L.BF16), st))

    def dg_red():
        L.check(raw_.favae_conv_igemm_red_groups(L.ptr(dy), L.ptr(wt), None, L.ptr(dx), _i(B), _i(Ho), _i(Ho), _i(Co), _i(Hi), _i(Hi), _i(Ci), _i(3), _i(2),
                                                 _i(1), _i(sc_b), _i(0), _i(L.BF16), L.ptr(bins), _i(nb(64 * Hi * Hi)), _i(2), L.ptr(actin), L.ptr(rawin),
                                                 L.ptr(mean), L.ptr(invstd), None, _i(64), ctypes.byref(fused), st))

    flop = 2.0 * B * Ho * Ho * Co * 9 * Ci / (4 if kind == "t" else 1)
    mb_f = (x.numel() * 2 + out.numel() * 4) / 1e6
    mb_b = (dy.numel() * 2 + dx.numel() * 4) / 1e6
    t = {k: timeit(f) for k, f in (("fwd", fwd), ("fwd_red", fwd_red), ("dg", dg), ("dg_red", dg_red))}
    fr = fused.value
    for k in t:
        tot[k] += t[k]
    print("%-8s fwd %6.1f us (+sums %6.1f) %6.0f TF/s %5.2f TB/s | dgrad %6.1f us (+sums %6.1f, fused %d) %6.0f TF/s %5.2f TB/s"
          % (name, t["fwd"], t["fwd_red"], flop / t["fwd"] / 1e6, mb_f / t["fwd"], t["dg"], t["dg_red"], fr, flop / t["dg"] / 1e6, mb_b / t["dg"]), flush=True)
print("total: " + "  ".join("%s %.0f us" % kv for kv in tot.items()))
