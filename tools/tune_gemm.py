"""Sweep tile x split-K for every implicit-GEMM launch of one batch-64 training step (developer tool)."""
import sys
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
raw = lib.raw()
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
B = 64
dev = 'cuda'
def t(*s, d=dt): return torch.randn(*s, device=dev).to(d)
def timeit(fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
KC = ['64x64', '128x32']
PM = ['32x128', '128x32', '64x64']
SK = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64]
def sweep(name, fn, tiles, kc):
    raw.favae_debug_force_tile(-1, -1)
    base = timeit(lambda: fn(0))
    res = []
    for ti, tn in enumerate(tiles):
        raw.favae_debug_force_tile(ti if kc else -1, -1 if kc else ti)
        for sk in SK:
            try: res.append((timeit(lambda: fn(sk)), tn, sk))
            except Exception as e: pass
    raw.favae_debug_force_tile(-1, -1)
    res.sort()
    print('%-14s auto %6.1f | ' % (name, base) + '  '.join('%s/sk%d %.1f' % (tn, sk, us) for us, tn, sk in res[:5]), flush=True)
def conv(name, Cin, Cout, Hi, Ho, ks, stride, scatter):
    if Cin % 32: return
    x = t(B, Hi, Hi, Cin); w = t(Cout, ks * ks, Cin); out = torch.empty(B, Ho, Ho, Cout, device=dev)
    sweep(name, lambda sk: lib.conv_igemm(x, w, None, out, ks, stride, 1 if ks == 3 else 0, scatter, sk), KC, True)
def wgrad(name, Rc, rl, Cc, cl, Hs, Hl, ks, stride):
    r = t(B, Hl if rl else Hs, Hl if rl else Hs, Rc); c = t(B, Hl if cl else Hs, Hl if cl else Hs, Cc)
    dw = torch.zeros(Rc, ks * ks, Cc, device=dev)
    sweep(name, lambda sk: lib.conv_wgrad(r, rl, c, cl, dw, B, Hs, Hs, Hl, Hl, ks, stride, 1 if ks == 3 else 0, sk), PM, False)
enc = [(3, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8), (512, 1024, 4)]
for i, (ci, co, h) in enumerate(enc):
    conv('enc.%d fwd' % i, ci, co, h, h // 2, 3, 2, 0)
    if i: conv('enc.%d dgrad' % i, co, ci, h // 2, h, 3, 2, 1)
    wgrad('enc.%d wgrad' % i, co, 0, ci, 1, h // 2, h, 3, 2)
conv('fc fwd', 4096, 40, 1, 1, 1, 1, 0); wgrad('fc wgrad', 40, 0, 4096, 0, 1, 1, 1, 1)
conv('di dgrad', 4096, 20, 1, 1, 1, 1, 0); wgrad('di wgrad', 4096, 0, 20, 0, 1, 1, 1, 1)
dec = [(1024, 512, 2), (512, 256, 4), (256, 128, 8), (128, 64, 16), (64, 64, 32)]
for i, (ci, co, h) in enumerate(dec):
    conv('dec.%d fwd' % i, ci, co, h, 2 * h, 3, 2, 1)
    conv('dec.%d dgrad' % i, co, ci, 2 * h, h, 3, 2, 0)
    wgrad('dec.%d wgrad' % i, co, 1, ci, 0, h, 2 * h, 3, 2)
conv('out fwd', 64, 3, 64, 64, 3, 1, 0); wgrad('out wgrad', 3, 0, 64, 1, 64, 64, 3, 1)
