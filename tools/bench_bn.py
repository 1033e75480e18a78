# BatchNorm launches of the bf16 training step alone, at the layer shapes of the step (developer tool; GPU box).
# Reports the time per launch and the algorithmic bytes / time.  The Python call costs ~10 us per launch: for the small shapes read the
# GPU-side durations from `rocprofv3 --kernel-trace` of this script instead.
import ctypes, sys
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
i_ = ctypes.c_int
DT = 1   # FAVAE_BF16
def t(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
rawl = lib.raw()
for (R, C) in [(2048, 64), (64 * 64 * 64, 64), (64 * 64 * 64, 32), (64 * 32 * 32, 64), (64 * 16 * 16, 128), (64 * 8 * 8, 256), (64 * 32 * 32, 32), (64 * 16 * 16, 64)]:
    nb = 32
    raw = torch.randn(R, C, device='cuda')
    dact = torch.randn(R, C, device='cuda')
    act = torch.randn(R, C, device='cuda').to(torch.bfloat16)
    draw = torch.empty_like(act)
    bins = torch.randn(nb, 2, C, dtype=torch.float64, device='cuda').abs() * R / nb
    g = torch.rand(C, device='cuda') + 0.5; b = torch.randn(C, device='cuda')
    mean = torch.zeros(C, device='cuda'); inv = torch.ones(C, device='cuda')
    rm = torch.zeros(C, device='cuda'); rv = torch.ones(C, device='cuda')
    dg = torch.empty(C, device='cuda'); db = torch.empty(C, device='cuda'); dbias = torch.empty(C, device='cuda')
    scratch = torch.empty(1 << 20, dtype=torch.float64, device='cuda')
    def fwd():
        lib.check(rawl.favae_bn_train_fwd_bins(lib.ptr(raw), lib.ptr(bins), i_(nb), lib.ptr(g), lib.ptr(b), lib.ptr(act), lib.ptr(mean),
                  lib.ptr(inv), lib.ptr(rm), lib.ptr(rv), ctypes.c_longlong(R), i_(C), ctypes.c_float(1e-5), ctypes.c_float(0.1), i_(DT), lib.stream_ptr()))
    def bwd():
        lib.check(rawl.favae_bn_lrelu_bwd_bins(lib.ptr(dact), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(inv), lib.ptr(g), lib.ptr(bins),
                  i_(nb), lib.ptr(draw), lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), ctypes.c_longlong(R), i_(C), lib.ptr(scratch),
                  ctypes.c_size_t(scratch.numel()), i_(DT), lib.stream_ptr()))
    big = torch.empty(256 * C + 4096, dtype=torch.float64, device='cuda')
    def bwd_defer():   # FAVAE_DBIAS_DEFER: per-slab column sums left in the buffer (kernel name ... BnBwdApplyF<..., true>)
        lib.check(rawl.favae_bn_lrelu_bwd_bins(lib.ptr(dact), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(inv), lib.ptr(g), lib.ptr(bins),
                  i_(nb), lib.ptr(draw), lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), ctypes.c_longlong(R), i_(C), lib.ptr(big),
                  ctypes.c_size_t(big.numel()), i_(DT | 0x80), lib.stream_ptr()))
    t(bwd_defer)
    tf, tb = t(fwd), t(bwd)
    print('R %7d C %3d  fwd %5.1f us %5.2f TB/s   bwd %5.1f us %5.2f TB/s' % (R, C, tf, R * C * 6 / tf / 1e6, tb, R * C * 12 / tb / 1e6), flush=True)
