"""GPU parity of the float64 factor-analysis kernels against the oracle's NumPy expressions
(oracle/fa_numpy.py, which restates condition_a_VAE/favae.py:904-1229).  Tolerance 1e-11 relative (fp64 re-association)."""
import numpy as np
import pytest
import torch

from oracle import fa_numpy as fo

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _fa():
    from favae_hip import fa
    return fa


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N,da,db,ones", [(512, 20, 10, True), (2048, 40, 100, True), (20001, 40, 50, True), (777, 50, 50, True),
                                           (1000, 40, 0, True), (4096, 130, 200, False), (3, 5, 7, True)])
def test_atb_statistics(N, da, db, ones):
    fa = _fa()
    rs = np.random.RandomState(N)
    A = rs.randn(N, da)
    B = rs.randn(N, db) if db else None
    sc = fa.Scratch("cuda")
    C = fa.atb(dev(A), None if B is None else dev(B), ones, sc).cpu().numpy()
    Bp = np.hstack([B, np.ones((N, 1))]) if (ones and db) else (np.ones((N, 1)) if not db else B)
    assert _rel(C, A.T @ Bp) < 1e-12
    # empty-ragged tail + reproducibility: same call twice is bitwise identical
    C2 = fa.atb(dev(A), None if B is None else dev(B), ones, sc).cpu().numpy()
    assert np.array_equal(C, C2)


@pytest.mark.parametrize("N,da,ones", [(20001, 50, True), (777, 64, True), (3000, 20, False), (4096, 130, True), (512, 65, True), (5, 3, True)])
def test_atb_of_a_matrix_with_itself(N, da, ones):
    """Z^T [Z | 1] (cA/favae.py:989, 1070): the same buffer as both operands takes the triangular path (rows loaded once, the tiles
    below the diagonal written as transposes); bitwise the result of the full product (a copy of Z as second operand)."""
    fa = _fa()
    A = np.random.RandomState(N + da).randn(N, da)
    sc = fa.Scratch("cuda")
    Ad = dev(A)
    C = fa.atb(Ad, Ad, ones, sc).cpu().numpy()
    Bp = np.hstack([A, np.ones((N, 1))]) if ones else A
    assert _rel(C, A.T @ Bp) < 1e-12
    assert np.array_equal(C[:, :da], C[:, :da].T)
    C2 = fa.atb(Ad, Ad.clone(), ones, sc).cpu().numpy()
    assert np.array_equal(C, C2)


@pytest.mark.parametrize("N,d,K", [(512, 20, 10), (2048, 40, 100), (20001, 50, 20), (100, 200, 40), (65, 3, 200), (300, 200, 512),
                                    (20001, 40, 50), (1000, 70, 51), (333, 130, 33), (64, 1, 1)])
def test_rows_gemm(N, d, K):
    fa = _fa()
    rs = np.random.RandomState(d)
    X, P, c, base = rs.randn(N, d), rs.randn(d, K), rs.randn(K), rs.randn(N, K)
    out = torch.zeros(N, K, dtype=F64, device="cuda")
    fa.rows_gemm(dev(X), dev(P), dev(c), 0.7, dev(base), 0.3, out)
    assert _rel(out.cpu().numpy(), 0.7 * (X @ P + c) + 0.3 * base) < 1e-12
    fa.rows_gemm(dev(X), dev(P), None, 2.0, None, 0.0, out, accumulate=True)
    assert _rel(out.cpu().numpy(), 0.7 * (X @ P + c) + 0.3 * base + 2.0 * X @ P) < 1e-12


@pytest.mark.parametrize("K", [1, 4, 10, 15, 16, 17, 31, 33, 50, 100, 130, 200, 201, 202, 260])
def test_chol_inverse_matches_my_inverse(K):
    fa = _fa()
    rs = np.random.RandomState(K)
    G = rs.randn(K + 5, K)
    A = np.eye(K) + G.T @ G
    sc = fa.Scratch("cuda")
    inv, ld_f, ld_o, ok = fa.chol_inv(dev(A), sc)
    assert ok
    assert _rel(inv.cpu().numpy(), fo.my_inverse(A)) < 1e-10
    assert abs(ld_f - np.linalg.slogdet(A)[1]) < 1e-9 * max(1, abs(ld_f))
    # update_w form: (1/tau) ss^T o (ss^T o ZZ + I/tau)^-1 == (diag(alpha) + tau ZZ)^-1
    alpha, tau = rs.rand(K) + 0.5, 1.9
    s = 1 / np.sqrt(alpha)
    inv2, _, ld2, ok = fa.chol_inv(dev(A), sc, s=dev(s), diag_add=1 / tau, out_scale=1 / tau)
    ref = np.linalg.inv(np.diag(alpha) + tau * A)
    assert ok and _rel(inv2.cpu().numpy(), ref) < 1e-10
    assert abs(ld2 - np.linalg.slogdet(ref)[1]) < 1e-9 * max(1, abs(ld2))


@pytest.mark.parametrize("K", [200, 50, 7])
def test_chol_inverse_batch_of_five_in_one_launch(K):
    """One update_Z of variant A with two views (cA/favae.py:970-980) at the stress size: Sigma_Z and the two predictive
    covariances -- whose `aux` is carried over from view to view -- plus the two W covariances of update_w (:1062-1065) with
    s = alpha.mean^-1/2 taken from alpha.b on the device: five K x K inverses, one launch, the sums formed while loading.
    One matrix of a second batch is not SPD: its output is NaN and flagged, the others are untouched (asynchronous: the
    flags are read once, afterwards)."""
    fa = _fa()
    rs = np.random.RandomState(K)
    G0, G1, Gz = rs.randn(K + 9, K), rs.randn(K + 3, K), rs.randn(2 * K, K)
    WW = [G0.T @ G0 / K, G1.T @ G1 / K]
    ZZ = Gz.T @ Gz
    taus = [1.7, 0.6]
    alpha_a, alpha_b = [0.52, 0.51], [rs.rand(K) + 0.2, rs.rand(K) + 0.3]
    I = np.eye(K)
    wwd, zzd, Id = [dev(w) for w in WW], dev(ZZ), dev(I)
    sc = fa.Scratch("cuda")
    mats = [dict(terms=[(1.0, Id), (taus[0], wwd[0]), (taus[1], wwd[1])]),                       # :970-972
            dict(terms=[(1.0, Id), (taus[1], wwd[1])]),                                         # m = 0: aux = I + others
            dict(terms=[(1.0, Id), (taus[1], wwd[1]), (taus[0], wwd[0])])]                      # m = 1: aux carried over (:979)
    for m in range(2):                                                                          # :1062-1065
        mats.append(dict(terms=[(1.0, zzd)], s=dev(alpha_b[m]), s_mul=1.0 / alpha_a[m], s_sqrt=True, diag_add=1 / taus[m], out_scale=1 / taus[m]))
    outs, logdet, info = fa.chol_inv_batched(mats, K, sc)
    aux = [I + taus[0] * WW[0] + taus[1] * WW[1], I + taus[1] * WW[1], I + taus[1] * WW[1] + taus[0] * WW[0]]
    refs = [fo.my_inverse(a) for a in aux]
    for m in range(2):
        refs.append(np.linalg.inv(np.diag(alpha_a[m] / alpha_b[m]) + taus[m] * ZZ))
    assert info.cpu().tolist() == [0] * 5
    ld = logdet.cpu().numpy()
    for b in range(5):
        assert _rel(outs[b].cpu().numpy(), refs[b]) < 1e-10, b
        assert abs(ld[b, 1] - np.linalg.slogdet(refs[b])[1]) < 1e-9 * max(1, abs(ld[b, 1])), b
    for b in range(3):
        assert abs(ld[b, 0] - np.linalg.slogdet(aux[b])[1]) < 1e-9 * max(1, abs(ld[b, 0]))
    bad = np.array(aux[1])
    bad[K // 2, K // 2] = -1.0
    keep = outs[0].clone()
    outs2, _, info2 = fa.chol_inv_batched([dict(terms=[(1.0, dev(aux[0]))]), dict(terms=[(1.0, dev(bad))]), dict(terms=[(0.5, dev(aux[2])), (0.5, dev(aux[2]))])],
                                          K, sc)
    assert info2.cpu().tolist() == [0, 1, 0]
    assert torch.isnan(outs2[1]).all() and _rel(outs2[0].cpu().numpy(), keep.cpu().numpy()) < 1e-12 and _rel(outs2[2].cpu().numpy(), refs[2]) < 1e-10


def test_chol_inverse_flags_non_spd_like_the_reference():
    fa = _fa()
    A = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert np.isnan(fo.my_inverse(A))            # reference path: exception -> NaN -> "not updated"
    inv, _, _, ok = fa.chol_inv(dev(A), fa.Scratch("cuda"))
    assert not ok and torch.isnan(inv).all()


@pytest.mark.parametrize("K,Kp", [(3, 1), (30, 1), (100, 1), (160, 1), (30, 2), (100, 7)])
def test_slogdet_of_depruned_covariance(K, Kp):
    """depruning (cA/favae.py:518-537) pads the pruned covariances with 1e-15: one pruned factor gives a regular
    matrix, two or more give identical rows, i.e. sign 0 / -inf, which HGauss then discards (:1619-1622)."""
    fa = _fa()
    rs = np.random.RandomState(K)
    Kk = K - Kp
    G = rs.randn(K, Kk)
    A = 1e-15 * np.ones((K, K))
    A[:Kk, :Kk] = np.linalg.inv(np.eye(Kk) + G.T @ G)
    sgn, lad = fa.slogdet(dev(A), fa.Scratch("cuda"))
    rs_, rl = np.linalg.slogdet(A)
    assert sgn == rs_
    if np.isinf(rl):
        assert lad == rl
    else:
        assert abs(lad - rl) < 1e-9 * abs(rl)
    B = rs.randn(K, K)                                   # general matrix with a negative determinant sign possible
    sgn, lad = fa.slogdet(dev(B), fa.Scratch("cuda"))
    rs_, rl = np.linalg.slogdet(B)
    assert sgn == rs_ and abs(lad - rl) < 1e-9 * max(1.0, abs(rl))


def test_multilabel_update_matches_oracle_expressions():
    fa = _fa()
    rs = np.random.RandomState(4)
    N, d, tau = 3001, 40, 1.3
    t = (rs.rand(N, d) > 0.5).astype(np.float64)
    zwb = rs.randn(N, d)
    xi = np.abs(rs.randn(N, d)) + 0.1
    cov = (tau + 2 * fo.lambda_func(xi)) ** (-1)
    mean = (t - 0.5 + tau * zwb) * cov
    xin = np.sqrt(cov + mean ** 2)
    e_ref = np.sum(np.log(fo.sigmoid(xin)) + mean * t - 0.5 * (mean + xin))
    xi_d, m_d, c_d = dev(xi), torch.zeros(N, d, dtype=F64, device="cuda"), torch.zeros(N, d, dtype=F64, device="cuda")
    sums = fa.mult_update(dev(t), dev(zwb), xi_d, m_d, c_d, tau, fa.Scratch("cuda"))
    assert _rel(c_d.cpu().numpy(), cov) < 1e-13 and _rel(m_d.cpu().numpy(), mean) < 1e-13 and _rel(xi_d.cpu().numpy(), xin) < 1e-13
    assert abs(sums[0] - np.sum(np.log(cov))) < 1e-11 * abs(sums[0])
    assert abs(sums[1] - e_ref) < 1e-11 * abs(e_ref)


def test_w_b_tau_alpha_updates_from_statistics():
    fa = _fa()
    rs = np.random.RandomState(9)
    N, d, K = 700, 12, 7
    X, Z = rs.randn(N, d) + 0.3, rs.randn(N, K)
    b = rs.randn(1, d) * 0.2
    G = rs.randn(K, K)
    wcov = np.linalg.inv(np.eye(K) + G.T @ G)
    tau, a0, b0 = 1.7, 1e-14, 1e-14
    sc = fa.Scratch("cuda")
    Xd, Zd = dev(X), dev(Z)
    xtz1 = fa.atb(Xd, Zd, True, sc)
    ztz1 = fa.atb(Zd, Zd, True, sc)
    xtx1 = fa.atb(Xd, Xd, True, sc)
    zsum, xsum = ztz1[:, K].contiguous(), xtx1[:, d].contiguous()
    wmean, wprodT = torch.zeros(d, K, dtype=F64, device="cuda"), torch.zeros(K, K, dtype=F64, device="cuda")
    fa.w_update(xtz1, zsum, dev(b.ravel()), dev(wcov), tau, wmean, wprodT, sc)
    W_ref = tau * np.linalg.multi_dot([(X - b).T, Z, wcov])
    assert _rel(wmean.cpu().numpy(), W_ref) < 1e-11
    assert _rel(wprodT.cpu().numpy(), W_ref.T @ W_ref + d * wcov) < 1e-11
    bmean, scal = torch.zeros(d, dtype=F64, device="cuda"), torch.zeros(4, dtype=F64, device="cuda")
    fa.b_update(xsum, zsum, wmean, tau, N, bmean, scal)
    bc = 1 / (1 + N * tau)
    b_ref = tau * np.sum(X - Z @ W_ref.T, axis=0) * bc
    assert _rel(bmean.cpu().numpy(), b_ref) < 1e-10
    bprodT = np.sum(b_ref ** 2) + d * bc
    assert abs(scal[1].item() - bprodT) < 1e-12 * bprodT
    zprodT = Z.T @ Z + N * np.eye(K) * 0.01
    xprodT_tr = np.trace(X.T @ X)
    fa.tau_update(xsum, zsum, xtz1, xprodT_tr, dev(zprodT), wprodT, wmean, bmean, bprodT, a0, b0, N, scal)
    Wp = W_ref.T @ W_ref + d * wcov
    s = (xprodT_tr + np.trace(Wp @ zprodT) - 2 * np.trace(np.linalg.multi_dot([W_ref, Z.T, X]))
         + 2 * np.sum(np.linalg.multi_dot([Z, W_ref.T, b_ref[:, None]])) - 2 * np.sum(X @ b_ref[:, None]) + N * bprodT)
    assert abs(scal[3].item() - (b0 + 0.5 * s) / (d * N)) < 1e-9 * abs((b0 + 0.5 * s) / (d * N))
    assert abs(scal[2].item() - (a0 + 0.5 * d * N) / (d * N)) < 1e-15
    ab = torch.zeros(K, dtype=F64, device="cuda")
    fa.alpha_update(wprodT, b0, d, ab)
    assert _rel(ab.cpu().numpy(), (b0 + 0.5 * np.diag(Wp)) / d) < 1e-12
    aux = fa.z_aux([wprodT, wprodT], [tau, 0.5], K, "cuda")
    assert _rel(aux.cpu().numpy(), np.eye(K) + (tau + 0.5) * Wp) < 1e-12
    sel = torch.tensor([0, 2, 5], dtype=torch.int32, device="cuda")
    out = torch.zeros(N, 3, dtype=F64, device="cuda")
    fa.gather_cols(Zd, sel, out)
    assert np.array_equal(out.cpu().numpy(), Z[:, [0, 2, 5]])
