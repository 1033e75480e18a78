import sys
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
P = 65536
f = torch.randn(P, 32, device='cuda').to(torch.bfloat16); f[:, 27:] = 0
w = torch.randn(64, 27, device='cuda').to(torch.bfloat16)
b = torch.randn(64, device='cuda')
out = torch.empty(P, 64, device='cuda')
bins = torch.zeros(64, 2, 64, dtype=torch.float64, device='cuda')
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
print('folded fwd + sums %.1f us' % t(lambda: lib.conv_folded_fwd(f, w, b, out, bins)))
print('folded fwd        %.1f us' % t(lambda: lib.conv_folded_fwd(f, w, b, out)))
