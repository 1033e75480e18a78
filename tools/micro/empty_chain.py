# Floor of a dependent launch on this box: back-to-back empty-ish torch kernels on one stream (developer tool)
import torch
x = torch.zeros(64, device='cuda')
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
print('x.add_(1) on 64 floats: %.2f us per launch' % t(lambda: x.add_(1)))
