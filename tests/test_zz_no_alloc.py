"""favae_fa_atb_static (the entry point that allocates nothing: slab partials in a file-scope device array) against
favae_fa_atb (caller-owned scratch) through the C ABI: same kernels, same geometry, so the results are bit-identical.
Two calls in a row, the second with a smaller input (the static array keeps the first call's partials), and a size
beyond the array's maximum is rejected with a clear error."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _call_static(raw, L, A, B, ones, C):
    N, da = A.shape
    db = 0 if B is None else B.shape[1]
    return raw.favae_fa_atb_static(L.ptr(A), ctypes.c_int(A.stride(0)), ctypes.c_int(da), L.ptr(B),
                                   ctypes.c_int(0 if B is None else B.stride(0)), ctypes.c_int(db), ctypes.c_int(1 if ones else 0),
                                   ctypes.c_longlong(N), L.ptr(C), L.stream_ptr())


def test_static_scratch_entry_point_equals_the_allocating_one():
    from favae_hip import fa
    from favae_hip import lib as L
    raw = L.raw()
    sc = fa.Scratch("cuda")
    rs = np.random.RandomState(7)
    # (N, da, db, ones): a large call first, then smaller ones that leave most of the first call's partials in place
    for N, da, db, ones in [(50000, 40, 50, True), (777, 20, 10, True), (3, 5, 7, False), (4096, 130, 200, True), (1000, 40, 0, True)]:
        A = torch.from_numpy(rs.randn(N, da)).cuda()
        B = torch.from_numpy(rs.randn(N, db)).cuda() if db else None
        ref = fa.atb(A, B, ones, sc)
        out = torch.full_like(ref, float("nan"))
        L.check(_call_static(raw, L, A, B, ones, out))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), (N, da, db, ones)
        Bp = np.ones((N, 1)) if B is None else (np.hstack([B.cpu().numpy(), np.ones((N, 1))]) if ones else B.cpu().numpy())
        want = A.cpu().numpy().T @ Bp
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-12 * np.abs(want).max()


def test_static_scratch_entry_point_rejects_sizes_beyond_its_array():
    from favae_hip import lib as L
    raw = L.raw()
    A = torch.zeros(4, 2100, dtype=F64, device="cuda")
    B = torch.zeros(4, 2100, dtype=F64, device="cuda")
    C = torch.zeros(2100, 2100, dtype=F64, device="cuda")
    rc = _call_static(raw, L, A, B, False, C)      # 2100 * 2100 > 4194304 doubles
    assert rc < 0 and "exceeds the static scratch" in L.last_error()
