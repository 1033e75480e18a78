"""Fused-input convolution (favae_conv_igemm_bn with in_ab) against the plain launch on the five fused transitions of a batch-64 step (developer tool)."""
import sys
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
B = 64
def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, Cin, Cout, Hi, scatter in [("enc.1", 64, 128, 32, 0), ("enc.2", 128, 256, 16, 0), ("enc.3", 256, 512, 8, 0), ("dec.3", 128, 64, 16, 1), ("final0", 64, 64, 32, 1)]:
    Ho = Hi * 2 if scatter else Hi // 2
    raw = torch.randn(B, Hi, Hi, Cin, device='cuda')
    act = raw.to(torch.bfloat16)
    w = torch.randn(Cout, 9, Cin, device='cuda').to(torch.bfloat16)
    out = torch.zeros(B, Ho, Ho, Cout, device='cuda')
    ab = torch.randn(2 * Cin, device='cuda')
    bins = torch.zeros(32, 2, Cout, dtype=torch.float64, device='cuda')
    sk = -1
    t0 = timeit(lambda: lib.conv_igemm(act, w, None, out, 3, 2, 1, scatter, sk))
    t1 = timeit(lambda: lib.conv_igemm_bn(raw, ab, w, None, out, 3, 2, 1, scatter, splitk=sk))
    t2 = timeit(lambda: lib.conv_igemm_red(act, w, None, out, 3, 2, 1, scatter, bins, 1, splitk=sk))
    apply_t = timeit(lambda: lib.bn_lrelu_fwd_ab(raw, ab, act))
    print('%-7s plain %6.1f us   +epilogue sums %6.1f us   fused input %6.1f us   (apply launch %5.1f us)' % (name, t0, t2, t1, apply_t))
