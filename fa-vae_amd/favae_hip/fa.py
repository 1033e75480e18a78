"""ctypes wrappers of the float64 factor-analysis kernels (csrc/fa.hip).  Tensors are device buffers only."""
import ctypes

import torch

from . import dist
from . import lib as L

_raw = L.raw()
_raw.favae_fa_atb_scratch_doubles.restype = ctypes.c_longlong
_i, _ll, _d, _sz = ctypes.c_int, ctypes.c_longlong, ctypes.c_double, ctypes.c_size_t
F64 = torch.float64


class Scratch:
    """Grow-only fp64 scratch shared by the reductions of one device."""

    def __init__(self, dev):
        self.dev = dev
        self.buf = torch.empty(1 << 16, dtype=F64, device=dev)
        self.work = torch.empty(1 << 16, dtype=F64, device=dev)
        self.info = torch.zeros(4, dtype=torch.int32, device=dev)
        self.pair = torch.zeros(4, dtype=F64, device=dev)

    def need(self, n):
        if self.buf.numel() < n:
            self.buf = torch.empty(int(n * 1.25), dtype=F64, device=self.dev)
        return self.buf

    def need_work(self, n):
        if self.work.numel() < n:
            self.work = torch.empty(int(n * 1.25), dtype=F64, device=self.dev)
        return self.work


def atb(A, B, ones, sc, out=None):
    """C[da][db+ones] = A^T [B | 1]; A [N, da], B [N, db] or None."""
    N, da = A.shape
    db = 0 if B is None else B.shape[1]
    dbp = db + (1 if ones else 0)
    C = out if out is not None else torch.empty(da, dbp, dtype=F64, device=A.device)
    n = _raw.favae_fa_atb_scratch_doubles(_ll(N), _i(da), _i(db), _i(1 if ones else 0))
    s = sc.need(n)
    L.check(_raw.favae_fa_atb(L.ptr(A), _i(A.stride(0)), _i(da), L.ptr(B), _i(0 if B is None else B.stride(0)), _i(db),
                              _i(1 if ones else 0), _ll(N), L.ptr(C), L.ptr(s), _sz(s.numel()), L.stream_ptr()))
    return C


def rows_gemm(X, P, c, alpha, base, beta, out, accumulate=False):
    """out = alpha (X P + c) + beta base  (accumulate: out += alpha X P)."""
    N, d = X.shape
    K = P.shape[1]
    L.check(_raw.favae_fa_rows_gemm(L.ptr(X), _i(X.stride(0)), _i(d), L.ptr(P), _i(P.stride(0)), _i(K), L.ptr(c), _d(alpha),
                                    L.ptr(base), _i(0 if base is None else base.stride(0)), _d(beta), L.ptr(out),
                                    _i(out.stride(0)), _ll(N), _i(1 if accumulate else 0), L.stream_ptr()))
    return out


def mult_update(t, zwb, xi, mean, cov, tau, sc, sync=True):
    s = sc.need(4096)
    L.check(_raw.favae_fa_mult_update(L.ptr(t), L.ptr(zwb), L.ptr(xi), L.ptr(mean), L.ptr(cov), _d(tau), _ll(t.numel()),
                                      L.ptr(sc.pair), L.ptr(s), _sz(s.numel()), L.stream_ptr()))
    dist.sum_(sc.pair)                          # the two sums run over all shards
    if not sync:
        return sc.pair[:2].clone()              # device: (sum log cov, bound term), read when the bound is assembled
    return sc.pair[:2].cpu().numpy().copy()     # (sum log cov, bound term)


def mult_ss_update(mean, mask_u8, t, sc):
    """update_t + write-back of the unobserved labels; returns (sum log tS.cov, change of the bound term sum X.mean * t.mean)."""
    s = sc.need(4096)
    L.check(_raw.favae_fa_mult_ss_update(L.ptr(mean), L.ptr(mask_u8), L.ptr(t), _ll(t.numel()), L.ptr(sc.pair), L.ptr(s),
                                         _sz(s.numel()), L.stream_ptr()))
    dist.sum_(sc.pair)
    return sc.pair[:2].cpu().numpy().copy()


def xi_init(mean, cov, xi):
    L.check(_raw.favae_fa_xi_init(L.ptr(mean), L.ptr(cov), L.ptr(xi), _ll(mean.numel()), L.stream_ptr()))


def mult_predict(m_x, scale, mean_t, var_t):
    N, d = m_x.shape
    L.check(_raw.favae_fa_mult_predict(L.ptr(m_x), L.ptr(scale), L.ptr(mean_t), L.ptr(var_t), _ll(N), _i(d), L.stream_ptr()))


def chol_inv_batched(mats, K, sc, lda=None):
    """favae_chol_inv_batched: up to 12 SPD inverses of size K in one asynchronous launch.  mats: dicts with
    terms = [(coef, [K, K] tensor), ...] (the matrix is their sum), optional s (tensor), s_mul, s_sqrt, diag_add, out_scale, out.
    Returns (outs, logdet [nb, 2], info [nb]) -- device tensors; nothing is synchronised."""
    nb = len(mats)
    dev = mats[0]["terms"][0][1].device
    A = (ctypes.c_void_p * (8 * nb))()
    coef = (ctypes.c_double * (8 * nb))()
    nterm = (ctypes.c_int * nb)()
    sv = (ctypes.c_void_p * nb)()
    s_mul = (ctypes.c_double * nb)()
    s_sqrt = (ctypes.c_int * nb)()
    dadd = (ctypes.c_double * nb)()
    oscale = (ctypes.c_double * nb)()
    outp = (ctypes.c_void_p * nb)()
    outs = []
    ld = None
    for b, m in enumerate(mats):
        nterm[b] = len(m["terms"])
        for t, (c, a) in enumerate(m["terms"]):
            assert a.dtype == F64 and a.shape[0] >= K and a.shape[1] >= K
            ld = a.stride(0) if ld is None else ld
            assert a.stride(0) == ld, "favae_chol_inv_batched: one leading dimension for all terms"
            A[8 * b + t] = a.data_ptr()
            coef[8 * b + t] = float(c)
        sb = m.get("s")
        sv[b] = None if sb is None else sb.data_ptr()
        s_mul[b] = float(m.get("s_mul", 1.0))
        s_sqrt[b] = 1 if m.get("s_sqrt") else 0
        dadd[b] = float(m.get("diag_add", 0.0))
        oscale[b] = float(m.get("out_scale", 1.0))
        o = m.get("out")
        if o is None:
            o = torch.empty(K, K, dtype=F64, device=dev)
        outs.append(o)
        outp[b] = o.data_ptr()
    logdet = torch.empty(nb, 2, dtype=F64, device=dev)
    info = torch.zeros(nb, dtype=torch.int32, device=dev)
    w = sc.need_work(2 * K * K + 2 * K) if K > 201 else None
    L.check(_raw.favae_chol_inv_batched(_i(nb), A, coef, nterm, _i(ld if lda is None else lda), _i(K), sv, s_mul, s_sqrt, dadd, oscale, outp, _i(K),
                                        L.ptr(logdet), L.ptr(info), L.ptr(w), _sz(0 if w is None else w.numel()), L.stream_ptr()))
    return outs, logdet, info


def chol_inv(A, sc, s=None, diag_add=0.0, out_scale=1.0):
    """(inverse [K,K] device tensor, logdet_factored, logdet_out, ok)."""
    K = A.shape[0]
    out = torch.empty(K, K, dtype=F64, device=A.device)
    w = sc.need_work(2 * K * K + 2 * K)
    rc = L.check(_raw.favae_chol_inv(L.ptr(A), _i(A.stride(0)), _i(K), L.ptr(s), _d(diag_add), _d(out_scale), L.ptr(out),
                                     _i(K), L.ptr(w), L.ptr(sc.pair), L.ptr(sc.info), L.stream_ptr()), allow_flags=True)
    ld = sc.pair[:2].cpu().numpy()
    return out, float(ld[0]), float(ld[1]), rc == 0


def slogdet(A, sc):
    K = A.shape[0]
    w = sc.need_work(K * K)
    L.check(_raw.favae_slogdet(L.ptr(A), _i(A.stride(0)), _i(K), L.ptr(w), L.ptr(sc.pair), L.stream_ptr()))
    r = sc.pair[:2].cpu().numpy()
    return float(r[0]), float(r[1])


def slogdet_into(A, out2):
    """(sign, log|det|) of A into the device pair out2, asynchronously (several matrices, one read-back by the caller)."""
    K = A.shape[0]
    w = torch.empty(K * K, dtype=F64, device=A.device)       # (own workspace: the calls of one bound are in flight together)
    L.check(_raw.favae_slogdet(L.ptr(A), _i(A.stride(0)), _i(K), L.ptr(w), L.ptr(out2), L.stream_ptr()))


def small_gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None):
    M = A.shape[1] if transA else A.shape[0]
    Kd = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    C = out if out is not None else torch.empty(M, N, dtype=F64, device=A.device)
    L.check(_raw.favae_small_gemm(_i(int(transA)), _i(int(transB)), _i(M), _i(N), _i(Kd), _d(alpha), L.ptr(A), _i(A.stride(0)),
                                  L.ptr(B), _i(B.stride(0)), _d(beta), L.ptr(C), _i(C.stride(0)), L.stream_ptr()))
    return C


def w_update(xtz1, zsum, b, wcov, tau, wmean, wprodT, sc, info=None, wcov_out=None, tmp=None):
    d, K1 = xtz1.shape
    K = K1 - 1
    tmp = sc.need_work(d * K) if tmp is None else tmp
    L.check(_raw.favae_fa_w_update(L.ptr(xtz1), L.ptr(zsum), L.ptr(b), L.ptr(wcov), _d(tau), _i(d), _i(K), L.ptr(wmean),
                                   L.ptr(wprodT), L.ptr(tmp), L.ptr(info), L.ptr(wcov_out), L.stream_ptr()))


def b_update(xsum, zsum, wmean, tau, N, bmean, scal):
    d, K = wmean.shape
    L.check(_raw.favae_fa_b_update(L.ptr(xsum), L.ptr(zsum), L.ptr(wmean), _d(tau), _d(float(N)), _i(d), _i(K), L.ptr(bmean),
                                   L.ptr(scal), L.stream_ptr()))


def tau_update(xsum, zsum, xtz1, trxx, zprodT, wprodT, wmean, bmean, bprodT, a0, b0, N, scal, bprodT_dev=None, xtx1_dev=None,
               covsum_dev=None):
    d, K = wmean.shape
    L.check(_raw.favae_fa_tau_update(L.ptr(xsum), L.ptr(zsum), L.ptr(xtz1), _d(trxx), L.ptr(zprodT), L.ptr(wprodT), L.ptr(wmean),
                                     L.ptr(bmean), _d(bprodT), _d(a0), _d(b0), _d(float(N)), _i(d), _i(K), L.ptr(scal),
                                     L.ptr(bprodT_dev), L.ptr(xtx1_dev), L.ptr(covsum_dev), L.stream_ptr()))


def alpha_update(wprodT, b0, d, alpha_b):
    K = wprodT.shape[0]
    L.check(_raw.favae_fa_alpha_update(L.ptr(wprodT), _d(b0), _i(d), _i(K), L.ptr(alpha_b), L.stream_ptr()))


def z_aux(wprodTs, coefs, K, dev):
    m = len(wprodTs)
    aux = torch.empty(K, K, dtype=F64, device=dev)
    ptrs = (ctypes.c_void_p * max(m, 1))(*[w.data_ptr() for w in wprodTs])
    cf = (ctypes.c_double * max(m, 1))(*[float(c) for c in coefs])
    L.check(_raw.favae_fa_z_aux(ptrs, cf, _i(m), _i(K), L.ptr(aux), L.stream_ptr()))
    return aux


def gather_cols(src, sel_dev, out):
    N, K = src.shape
    L.check(_raw.favae_gather_cols(L.ptr(src), _i(K), L.ptr(sel_dev), _i(sel_dev.numel()), L.ptr(out), _ll(N), L.stream_ptr()))
    return out


def axpby(a, x, b, y, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(_raw.favae_axpby(_ll(x.numel()), _d(a), L.ptr(x), _d(b), L.ptr(y), L.ptr(out), L.stream_ptr()))
    return out
