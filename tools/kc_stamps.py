"""Cycle stamps of one wave of the K-contiguous implicit GEMM (developer tool; needs a library built with -DFAVAE_KC_STAMPS:
   cd fa-vae_amd/csrc && hipcc ... -DFAVAE_KC_STAMPS -c igemm.hip && relink).  usage: python tools/kc_stamps.py [enc1|enc3|dec1]"""
import sys, ctypes
sys.path.insert(0, '/root/repo/fa-vae_amd')
import torch
from favae_hip import lib
which = sys.argv[1] if len(sys.argv) > 1 else 'enc1'
B = 64
Cin, Cout, Hi, Ho, scatter = {'enc1': (64, 128, 32, 16, 0), 'enc3': (256, 512, 8, 4, 0), 'dec1': (512, 256, 4, 8, 1)}[which]
x = torch.randn(B, Hi, Hi, Cin, device='cuda').to(torch.bfloat16)
w = torch.randn(Cout, 9, Cin, device='cuda').to(torch.bfloat16)
out = torch.empty(B, Ho, Ho, Cout, device='cuda')
for _ in range(5):
    lib.conv_igemm(x, w, None, out, 3, 2, 1, scatter, 0)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
raw = lib.raw()
raw.favae_debug_kc_stamps.argtypes = [ctypes.c_void_p]
assert raw.favae_debug_kc_stamps(buf) == 0
s = list(buf)
t0 = s[0]
names = {0: 'start', 1: 'addressing done', 2: 'first loads issued', 20: 'loop done', 21: 'end'}
for i in [0, 30, 31, 32, 33, 34] + list(range(1, 20)) + [20, 21]:
    if s[i]:
        print('%2d %-20s %8d cycles' % (i, names.get(i, 'chunk %d past barrier' % (i - 3)), s[i] - t0))

import numpy as np
blk = (ctypes.c_ulonglong * 16384)()
raw.favae_debug_kc_blocks.argtypes = [ctypes.c_void_p]
assert raw.favae_debug_kc_blocks(blk) == 0
b = np.array(list(blk), dtype=np.float64).reshape(-1, 2)
b = b[(b[:, 0] > 0) & (b[:, 1] > 0)]
t0 = b[:, 0].min()
st, en = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0          # us (100 MHz clock)
print('blocks %d: start us min %.2f median %.2f max %.2f; end us median %.2f max %.2f; lifetime us median %.2f max %.2f' % (
    len(b), st.min(), np.median(st), st.max(), np.median(en), en.max(), np.median(en - st), (en - st).max()))
