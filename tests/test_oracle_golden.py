"""Pin the CPU oracle (oracle/) against vectors produced by the real reference (oracle/make_goldens.py).

The reference ships no tests or fixtures; these goldens are outputs of the reference
itself, run in the build container with the shims of SURVEY.md section 8(c).  CPU only.
"""
import numpy as np
import pytest
import torch

from conftest import fa_inputs, img_inputs, ss_inputs, ss2_inputs, checksum
from oracle import fa_numpy as fa
from oracle import vae_torch as vt


def _close(a, b, rtol=1e-9, atol=1e-10):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def _check_q(model, g, rtol=1e-8):
    q = model.q_dist
    _close(model.L, g["L"], rtol=rtol)
    assert q.Kc == int(g["Kc"])
    _close(q.Z["mean"], g["Z_mean"], rtol=rtol, atol=1e-9)
    _close(q.Z["cov"], g["Z_cov"], rtol=rtol, atol=1e-12)
    for m in range(model.m):
        _close(q.W[m]["mean"], g["W%d_mean" % m], rtol=rtol, atol=1e-9)
        _close(q.b[m]["mean"], g["b%d_mean" % m], rtol=rtol, atol=1e-9)
        _close([q.tau[m]["a"], np.ravel(q.tau[m]["b"])[0]], g["tau%d" % m], rtol=rtol)
        _close(q.alpha[m]["b"], g["alpha%d_b" % m], rtol=rtol)
        _close(model.X[m]["mean"], g["X%d_mean" % m], rtol=rtol, atol=1e-9)
        _close(q.xi[m], g["xi%d" % m], rtol=rtol, atol=1e-9)


@pytest.mark.parametrize("variant,name", [("A", "fa_A.npz"), ("B", "fa_B.npz")])
def test_fa_only_matches_reference(golden, variant, name):
    g = golden(name)
    X0, lab = fa_inputs()
    np.random.seed(0)
    torch.manual_seed(0)
    model = fa.SSHIBAOracle(10, 1, variant=variant)
    model.fit(fa.struct_data(X0, "reg"), fa.struct_data(lab, "mult"), max_iter=4, pruning_crit=1e-3)
    _check_q(model, g)
    # SSHIBA.predict on the fitted model (cA/favae.py:1336-1483): multilabel view in, regression / multilabel view out
    gp = golden("predict_" + name)
    _, lab_t = fa_inputs(n=100, seed=2)
    p = model.predict([1], [0], fa.struct_data(lab_t, "mult"))
    _close(model.Z_mean, gp["Z_mean"], rtol=1e-8, atol=1e-10)
    _close(p["output_view0"]["mean_x"], gp["reg_mean"], rtol=1e-8, atol=1e-10)
    _close(p["output_view0"]["var_x"], gp["reg_var"], rtol=1e-8, atol=1e-12)
    p = model.predict([1], [1], fa.struct_data(lab_t, "mult"))
    _close(p["output_view1"]["mean_x"], gp["mult_mean"], rtol=1e-8, atol=1e-10)
    _close(p["output_view1"]["var_x"], gp["mult_var"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("variant,name", [("A", "fa_A_ss.npz"), ("B", "fa_B_ss.npz")])
def test_semi_supervised_reg_view_matches_reference(golden, variant, name):
    """'reg' view with unobserved entries (NaN): initial imputation draws, update_xs, masked covariance, X* entropy
    (cA/favae.py:236-257, 775-783, 1157-1174, 1651-1668)."""
    g = golden(name)
    X0, lab = ss_inputs()
    np.random.seed(0)
    torch.manual_seed(0)
    model = fa.SSHIBAOracle(10, 1, variant=variant)
    model.fit(fa.struct_data(X0, "reg"), fa.struct_data(lab, "mult"), max_iter=4, pruning_crit=1e-3)
    _check_q(model, g)
    _close(model.X[0]["cov"], g["X0_cov"], rtol=1e-8, atol=1e-12)
    _close(model.q_dist.XS[0]["mean"], g["XS0_mean"], rtol=1e-8, atol=1e-9)
    _close(model.X[0]["sumlogdet"], g["X0_sumlogdet"], rtol=1e-9)
    # a view with fewer rows than the others is padded with unobserved rows (:239-240); only b's initial covariance
    # differs from the NaN-row form (it uses n[m], :1986), so the two runs stay within a per cent of each other
    Xs, _ = fa_inputs()
    np.random.seed(0)
    a = fa.SSHIBAOracle(10, 1, variant=variant)
    a.fit(fa.struct_data(Xs[:384], "reg"), fa.struct_data(lab, "mult"), max_iter=2, pruning_crit=1e-3)
    assert a.X[0]["mean"].shape == (512, 20) and a.SS == [True, False] and int(a.SS_mask[0].sum()) == 128 * 20
    Xn = Xs.copy()
    Xn[384:] = np.nan
    np.random.seed(0)
    b = fa.SSHIBAOracle(10, 1, variant=variant)
    b.fit(fa.struct_data(Xn, "reg"), fa.struct_data(lab, "mult"), max_iter=2, pruning_crit=1e-3)
    _close(a.L, b.L, rtol=2e-2)


def test_semi_supervised_labels_match_reference(golden):
    """Unobserved labels in the multilabel view as well: initial draws (:276-289), update_t after every update_x/update_xi
    (:858-861, 1176-1195), t* entropy in the bound (:1655-1656)."""
    g = golden("fa_A_ss2.npz")
    X0, lab = ss2_inputs()
    np.random.seed(0)
    torch.manual_seed(0)
    model = fa.SSHIBAOracle(10, 1, variant="A")
    model.fit(fa.struct_data(X0, "reg"), fa.struct_data(lab, "mult"), max_iter=4, pruning_crit=1e-3)
    _check_q(model, g)
    assert model.SS == [True, True]
    _close(model.t[1]["mean"], g["t1_mean"], rtol=1e-8, atol=1e-10)
    _close(model.q_dist.tS[1]["sumlogdet"], g["tS1_sumlogdet"], rtol=1e-9)


def test_fa_pruning_matches_reference(golden):
    g = golden("fa_A_prune.npz")
    X0, lab = fa_inputs(rank=3)
    np.random.seed(0)
    model = fa.SSHIBAOracle(30, 1, variant="A")
    model.fit(fa.struct_data(X0, "reg"), fa.struct_data(lab, "mult"), max_iter=40, pruning_crit=1e-3)
    assert model.q_dist.Kc < 30
    _check_q(model, g, rtol=1e-6)


def test_vae_three_steps_match_reference(golden):
    g = golden("vae_step.npz")
    torch.manual_seed(0)
    np.random.seed(0)
    P, B = vt.init_state(20, 3)
    names = list(P.keys())
    opt = torch.optim.Adam([P[k] for k in names], 1e-3)
    Bn = 16
    img, _ = img_inputs(Bn)
    prior = torch.from_numpy(np.random.RandomState(7).randn(Bn, 20) * 0.5)
    tau = torch.Tensor(np.asarray(np.array([1.7])))
    images = torch.tensor(img)
    for step in range(3):
        opt.zero_grad()
        torch.manual_seed(100 + step)
        loss, rec_err, kl, mu, std, z, rec = vt.loss_terms(P, B, images, prior, tau, 0.01, 1)
        loss.backward()
        _close([loss.item(), rec_err.item(), kl.item()], g["s%d_loss" % step], rtol=1e-6)
        _close(mu.detach().numpy(), g["s%d_mu" % step], rtol=1e-5, atol=1e-6)
        _close(std.detach().numpy(), g["s%d_std" % step], rtol=1e-5, atol=1e-6)
        _close(checksum(rec.detach().numpy()), g["s%d_rec" % step], rtol=1e-5, atol=1e-6)
        for k in names:
            _close(checksum(P[k].grad.numpy()), g["s%d_g_%s" % (step, k)], rtol=1e-4, atol=1e-5)
        opt.step()
        for k in names:
            _close(checksum(P[k].detach().numpy()), g["s%d_p_%s" % (step, k)], rtol=1e-5, atol=1e-6)
        for k in B:
            _close(checksum(B[k].numpy()), g["s%d_p_%s" % (step, k)], rtol=1e-5, atol=1e-6)


def test_vae_trainloop_update_x_reconstruction_match_reference(golden):
    g = golden("vae_epoch.npz")
    torch.manual_seed(0)
    np.random.seed(0)
    vae = vt.OracleImgVAE(dimx=8, channels=3, lr=1e-3, dataset="g")
    N = 24
    img, _ = img_inputs(N)
    rs = np.random.RandomState(3)
    Z = dict(mean=rs.randn(N, 4), cov=np.eye(4))
    W = dict(mean=rs.randn(8, 4) * 0.3, cov=np.eye(4))
    b = dict(mean=rs.randn(1, 8) * 0.1, cov=np.eye(8))
    keep = vae.trainloop(img=img, Z=Z, W=W, b=b, tau=np.array([2.5]), epochs=3, favae=True, beta=1)
    mu, var = vae.update_x(img=img)
    rec = vae.reconstruction(mu, var)
    assert bool(keep) == bool(g["keep"])
    _close(vae.elbo_training, g["elbo"], rtol=1e-6)
    _close(vae.reconstruc_during_training, g["rec_loss"], rtol=1e-6)
    _close(vae.KL_QandP, g["kl"], rtol=1e-6)
    _close(mu, g["mu"], rtol=1e-4, atol=1e-5)
    _close(var, g["var"], rtol=1e-4, atol=1e-7)
    _close(checksum(rec), g["rec"], rtol=1e-4, atol=1e-5)
    for k, p in vae.state_dict().items():
        _close(checksum(p.numpy()), g["p_" + k], rtol=1e-4, atol=1e-6)


def test_full_variant_A_fit_matches_reference(golden):
    """One image view (z=8) + one multilabel view, N=32, Kc=4, two SSHIBA iterations (50 + 5 epochs)."""
    g = golden("favae_A.npz")
    N = 32
    img, lab = img_inputs(N)
    np.random.seed(0)
    torch.manual_seed(0)
    model = fa.SSHIBAOracle(4, 1, variant="A", vae_factory=lambda **kw: vt.OracleImgVAE(**kw))
    v0 = fa.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g")
    v1 = fa.struct_data(lab, "mult")
    model.fit(v0, v1, max_iter=2, pruning_crit=1e-3)
    _close(model.L, g["L"], rtol=1e-5)
    _close(model.img_vae[0].elbo_training, g["elbo_last"], rtol=1e-5)
    _close(model.total_elbo[0], g["total_elbo0"], rtol=1e-5)
    _close(model.q_dist.Z["mean"], g["Z_mean"], rtol=1e-3, atol=1e-5)
    _close(model.X[0]["mean"], g["X0_mean"], rtol=1e-3, atol=1e-5)
    _close(model.q_dist.W[0]["mean"], g["W0_mean"], rtol=1e-3, atol=1e-6)
    # predict / predict_sameimage on the trained model, same seeds as oracle/make_goldens.py.  The tolerances are those
    # of the fit above (the predict arithmetic itself is pinned to 1e-8 by the FA-only test).
    gp = golden("favae_A_predict.npz")
    img_t, lab_t = img_inputs(12, seed=5)
    np.random.seed(11); torch.manual_seed(11)
    p = model.predict([0], [1], fa.struct_data(img_t, "img"))
    _close(model.Z_mean, gp["i2m_Z_mean"], rtol=2e-3, atol=1e-4)
    _close(p["output_view1"]["mean_x"], gp["i2m_mean"], rtol=2e-3, atol=1e-4)
    _close(p["output_view1"]["var_x"], gp["i2m_var"], rtol=2e-3, atol=1e-4)
    np.random.seed(12); torch.manual_seed(12)
    pred_img, zm, xc = model.predict([1], [0], fa.struct_data(lab_t, "mult"))
    assert tuple(pred_img.shape) == tuple(gp["m2i_img_shape"]) and xc == 0
    _close(zm, gp["m2i_Z_mean"], rtol=2e-3, atol=1e-4)
    _close(checksum(pred_img.numpy()), gp["m2i_img"], rtol=2e-3, atol=2e-3)
    np.random.seed(13); torch.manual_seed(13)
    pred_img, zm, xc = model.predict_sameimage([1], [0], fa.struct_data(lab_t, "mult"))
    _close(zm, gp["same_Z_mean"], rtol=2e-3, atol=1e-4)
    _close(checksum(pred_img), gp["same_img"], rtol=2e-3, atol=2e-3)
    _close(checksum(xc.numpy()), gp["same_xcart"], rtol=2e-3, atol=2e-3)


def test_full_variant_B_fit_matches_reference(golden):
    """domain_adaptation: two image views (z=8) + one multilabel view, N=16, Kc=4, two iterations (50 + 20 epochs per
    view, beta 0.1 / 0.5, sampled FA prior), then predict image view 0 -> image view 1."""
    g = golden("favae_B.npz")
    N = 16
    img0, lab = img_inputs(N)
    img1, _ = img_inputs(N, seed=4, zt=np.random.RandomState(1).randn(N, 5))
    lab = lab[:, :3]
    np.random.seed(0)
    torch.manual_seed(0)
    fac = lambda **kw: vt.OracleImgVAE(prior="sampled", force_beta_one=False, update_x_self_img=True, **kw)
    model = fa.SSHIBAOracle(4, 1, variant="B", vae_factory=fac, diagnostics=True)
    model.fit(fa.struct_data(img0, "img", latent_dim=8, lr=1e-3, dataset="g0"),
              fa.struct_data(img1, "img", latent_dim=8, lr=1e-3, dataset="g1"), fa.struct_data(lab, "mult"),
              max_iter=2, pruning_crit=1e-3)
    _close(model.L, g["L"], rtol=1e-4)
    for i in range(4):
        _close(model.total_elbo[i], g["total_elbo%d" % i], rtol=1e-5)
    _close(model.q_dist.Z["mean"], g["Z_mean"], rtol=2e-3, atol=1e-4)
    _close(model.X[1]["mean"], g["X1_mean"], rtol=2e-3, atol=1e-4)
    _close(model.q_dist.W[0]["mean"], g["W0_mean"], rtol=2e-3, atol=1e-5)
    img_t, _ = img_inputs(N, seed=5)
    np.random.seed(21); torch.manual_seed(21)
    pred_img, X_sample, Z_in, xcart = model.predict([0], [1], fa.struct_data(img_t, "img"))
    _close(X_sample, g["pred_X_sample"], rtol=2e-3, atol=1e-4)
    _close(Z_in["mean"], g["pred_Z_mean"], rtol=2e-3, atol=1e-4)
    _close(Z_in["cov"], g["pred_Z_cov"], rtol=2e-3, atol=1e-6)
    _close(checksum(pred_img), g["pred_img"], rtol=2e-3, atol=2e-3)
    _close(checksum(xcart.numpy()), g["pred_xcart"], rtol=2e-3, atol=2e-3)


def test_full_variant_C_fit_matches_reference(golden):
    """transfer_learning/favae.py + betaVAE.py: image + 'reg' (pretrained latents) + multilabel views as in
    TL/trainer_FAVAE.py:96, N=16, Kc=4, two iterations; the
    VAE keeps the N(0,I) prior (the coordinator never passes favae=True), beta follows the 0.1 / 0.5 schedule."""
    g = golden("favae_C.npz")
    N = 16
    img, lab = img_inputs(N)
    np.random.seed(0)
    torch.manual_seed(0)
    fac = lambda **kw: vt.OracleImgVAE(prior="sampled", force_beta_one=False, update_x_self_img=True, **kw)
    model = fa.SSHIBAOracle(4, 1, variant="C", vae_factory=fac)
    X0, _ = fa_inputs(n=N, seed=3)
    model.fit(fa.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g"), fa.struct_data(X0, "reg"),
              fa.struct_data(lab[:, :5], "mult"), max_iter=2, pruning_crit=1e-3)
    _close(model.L, g["L"], rtol=1e-4)
    for i in range(2):
        _close(model.total_elbo[i], g["total_elbo%d" % i], rtol=1e-5)
        _close(model.kl_q_p[i], g["kl%d" % i], rtol=1e-4)
    _close(model.X[0]["mean"], g["X0_mean"], rtol=2e-3, atol=1e-4)
    _close(model.q_dist.W[1]["mean"], g["W1_mean"], rtol=2e-3, atol=1e-5)
    _close(model.q_dist.W[2]["mean"], g["W2_mean"], rtol=2e-3, atol=1e-5)


def test_standalone_beta_vae_matches_reference(golden):
    """transfer_learning/betaVAE.py on its own: trainloop(epochs=3, beta=4) against N(0,I) (the second epoch is worse
    than the first, so restore_weights reloads the checkpoint), then update_x."""
    g = golden("vae_beta_epoch.npz")
    torch.manual_seed(0)
    np.random.seed(0)
    vae = vt.OracleImgVAE(dimx=8, channels=3, lr=1e-3, dataset="g", prior="sampled", force_beta_one=False, update_x_self_img=True)
    img, _ = img_inputs(24)
    keep = vae.trainloop(img=img, epochs=3, beta=4)
    assert bool(keep) == bool(g["keep"])
    _close(vae.elbo_training, g["elbo"], rtol=1e-5)
    _close(vae.KL_QandP, g["kl"], rtol=1e-4)
    mu, var = vae.update_x(img=img)
    _close(mu, g["mu"], rtol=1e-3, atol=1e-4)
    _close(var, g["var"], rtol=1e-3, atol=1e-6)
