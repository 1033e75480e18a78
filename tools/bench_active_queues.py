"""Does a one-wave kernel that stays resident on ANOTHER stream (what a polling stream-word wait is) slow the launches of the main stream?
Times a 1024-block convolution GEMM and a BatchNorm apply alone and with 1 / 3 sleeping one-wave kernels on other streams.
Run on the GPU box: python tools/bench_active_queues.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd")):
    sys.path.insert(0, p)
from favae_hip import lib as L  # noqa: E402

B = 64
shapes = [("enc.1 fwd 512 blocks", 64, 128, 32, 16, 0), ("dec.3 dgrad 1024 blocks", 64, 64, 32, 32, 1), ("enc.3 fwd split", 256, 512, 8, 4, 0)]
streams = [torch.cuda.Stream() for _ in range(3)]


def timeit(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, Ci, Co, Hi, Ho, s1 in shapes:
    x = torch.randn(B, Hi, Hi, Ci, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, 9, Ci, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.zeros(B, Ho, Ho, Co, device="cuda")

    def gemm():
        L.conv_igemm(x, w, None, out, 3, 1 if s1 else 2, 1, 0, 0)

    res = []
    for nsleep in (0, 1, 3):
        torch.cuda.synchronize()
        for i in range(nsleep):
            with torch.cuda.stream(streams[i]):
                torch.cuda._sleep(int(2.0e9))      # ~1 s: a resident single-wave kernel on another hardware queue
        res.append(timeit(gemm))
        torch.cuda.synchronize()
    print("%-26s alone %6.2f us | 1 resident wave elsewhere %6.2f us | 3: %6.2f us" % (name, res[0], res[1], res[2]), flush=True)
