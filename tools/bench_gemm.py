"""Per-layer timing of the implicit-GEMM launches of one batch-64 training step (developer tool)."""
import sys
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
import os, ctypes
B = int(os.environ.get('B', '64'))       # images per launch
if os.environ.get('FORCE_PM'):          # favae_debug_force_tile for the weight gradients (2: igemm_pm 64x64, 9: all-taps kernel)
    lib.raw().favae_debug_force_tile(ctypes.c_int(-1), ctypes.c_int(int(os.environ['FORCE_PM'])))
dev = 'cuda'
def t(*s, d=dt): return torch.randn(*s, device=dev).to(d)
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
rows = []
SPLITK = int(os.environ.get('SPLITK', '0'))    # 0: the library's choice, 1: unsplit reductions
ONLY = os.environ.get('ONLY')       # run only the layers whose name contains this (for rocprofv3 --pmc passes)
def conv(name, Cin, Cout, Hi, Ho, ks, stride, scatter, taps_eff):
    if ONLY and ONLY not in name: return
    x = t(B, Hi, Hi, Cin); w = t(Cout, ks * ks, Cin); out = torch.empty(B, Ho, Ho, Cout, device=dev)
    us = timeit(lambda: lib.conv_igemm(x, w, None, out, ks, stride, 1 if ks == 3 else 0, scatter, SPLITK))
    fl = 2.0 * B * Ho * Ho * Cout * taps_eff * Cin
    rows.append((name, us, fl / us / 1e6))
def wgrad(name, Rc, rl, Cc, cl, Hs, Hl, ks, stride):
    if ONLY and ONLY not in name: return
    r = t(B, Hl if rl else Hs, Hl if rl else Hs, Rc); c = t(B, Hl if cl else Hs, Hl if cl else Hs, Cc)
    dw = torch.zeros(Rc, ks * ks, Cc, device=dev)
    us = timeit(lambda: lib.conv_wgrad(r, rl, c, cl, dw, B, Hs, Hs, Hl, Hl, ks, stride, 1 if ks == 3 else 0, 0))
    fl = 2.0 * Rc * Cc * B * Hs * Hs * ks * ks
    rows.append((name, us, fl / us / 1e6))
enc = [(3, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8), (512, 1024, 4)]
for i, (ci, co, h) in enumerate(enc):
    conv('enc.%d fwd' % i, ci, co, h, h // 2, 3, 2, 0, 9)
    if i: conv('enc.%d dgrad' % i, co, ci, h // 2, h, 3, 2, 1, 2.25)
    wgrad('enc.%d wgrad' % i, co, 0, ci, 1, h // 2, h, 3, 2)
conv('fc fwd', 4096, 40, 1, 1, 1, 1, 0, 1); conv('fc dgrad', 40, 4096, 1, 1, 1, 1, 0, 1); wgrad('fc wgrad', 40, 0, 4096, 0, 1, 1, 1, 1)
conv('di fwd', 20, 4096, 1, 1, 1, 1, 0, 1); conv('di dgrad', 4096, 20, 1, 1, 1, 1, 0, 1); wgrad('di wgrad', 4096, 0, 20, 0, 1, 1, 1, 1)
dec = [(1024, 512, 2), (512, 256, 4), (256, 128, 8), (128, 64, 16), (64, 64, 32)]
for i, (ci, co, h) in enumerate(dec):
    conv('dec.%d fwd' % i, ci, co, h, 2 * h, 3, 2, 1, 2.25)
    conv('dec.%d dgrad' % i, co, ci, 2 * h, h, 3, 2, 0, 9)
    wgrad('dec.%d wgrad' % i, co, 1, ci, 0, h, 2 * h, 3, 2)
conv('out fwd', 64, 3, 64, 64, 3, 1, 0, 9); conv('out dgrad', 3, 64, 64, 64, 3, 1, 1, 9); wgrad('out wgrad', 3, 0, 64, 1, 64, 64, 3, 1)
tot = sum(r[1] for r in rows)
for n, us, tf in rows: print('%-14s %8.1f us %8.1f TF' % (n, us, tf))
print('total %.1f us' % tot)
