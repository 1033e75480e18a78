"""Generate tests/golden/*.npz by running the REAL reference in the build container.

TEST INFRASTRUCTURE ONLY.  Run once here (`python -m oracle.make_goldens`); the .npz files (inputs are
regenerated from seeds, outputs are stored) are committed.  /root/reference never travels.

Fixtures
  fa_A.npz / fa_B.npz     FA-only fits ('reg' d=20 + 'mult' d=40, N=512, Kc=10, prune=1) with
                          condition_a_VAE/favae.py (variant A) and domain_adaptation/favae.py (variant B).
  fa_A_prune.npz          variant A with Kc=30 on rank-3 data, 40 iterations: pruning removes factors.
  vae_step.npz            three optimiser steps of celeba_deepvanilla_vae.ImgVAE (z=20, batch 16) with an FA
                          prior: losses, mu/std, gradient and parameter checksums.
  vae_epoch.npz           ImgVAE.trainloop (N=24 => batches 16+8... uses batch 64 => one batch of 24) for
                          3 epochs + update_x + reconstruction: epoch metrics, X moments, checksums.
  fa_A_ss.npz / fa_B_ss.npz  the same FA-only fit with a semi-supervised 'reg' view (128 rows + scattered entries unobserved).
  fa_A_ss2.npz            ... and unobserved labels in the 'mult' view as well (rows 300-419, every 5th entry of label 7).
  predict_fa_A/B.npz      SSHIBA.predict after the FA-only fits: 'mult' view in -> 'reg' view out and -> 'mult' view out.
  favae_A_predict.npz     after the favae_A fit: predict img -> mult, mult -> img (decoder of a sampled latent),
                          predict_sameimage mult -> img (seeds set before each call).
  favae_B.npz             full variant-B (domain_adaptation) fit: two image views (z=8) + one 'mult' view, N=16, Kc=4,
                          2 iterations (50 + 20 epochs per view), then predict image view 0 -> image view 1.
  favae_C.npz             variant C (transfer_learning/favae.py + betaVAE.py: the VAE keeps its N(0,I) prior because the
                          coordinator never passes favae=True): image view (z=8) + 'reg' view (pretrained latents) + 'mult' view as in
                          TL/trainer_FAVAE.py, N=16, Kc=4, 2 iterations.
  vae_beta_epoch.npz      stand-alone beta-VAE (transfer_learning/betaVAE.py): trainloop(epochs=3, beta=4) + update_x.
  favae_A.npz             full variant-A fit, one image view (z=8) + one 'mult' view, N=32, Kc=4, 2 iterations
                          (50 + 5 epochs as the reference schedules them).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_harness as rh  # noqa: E402


def fa_inputs(n=512, rank=5, seed=1):
    rs = np.random.RandomState(seed)
    Zt = rs.randn(n, rank)
    X0 = Zt @ rs.randn(rank, 20) + 0.1 * rs.randn(n, 20)
    lab = (Zt @ rs.randn(rank, 40) > 0).astype(int)
    return X0, lab


def img_inputs(n, seed=1, zt=None):
    rs = np.random.RandomState(seed)
    Zt = rs.randn(n, 5) if zt is None else zt
    Pm = rs.randn(5, 3 * 64 * 64) * 0.7
    img = (1 / (1 + np.exp(-(Zt @ Pm)))).reshape(n, 3, 64, 64).astype(np.float32)
    lab = (Zt @ rs.randn(5, 40) > 0).astype(int)
    return img, lab


def checksum(t):
    a = np.asarray(t, dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, 64).astype(np.int64)
    return np.concatenate(([a.sum(), np.sqrt((a * a).sum())], a[idx]))


def dump_q(model, out, tag=""):
    q = model.q_dist
    out[tag + "L"] = np.array(model.L, dtype=np.float64)
    out[tag + "Kc"] = np.array(q.Kc)
    out[tag + "Z_mean"] = q.Z["mean"]
    out[tag + "Z_cov"] = q.Z["cov"]
    out[tag + "Z_prodT"] = q.Z["prodT"]
    for m in range(model.m):
        out[tag + "W%d_mean" % m] = q.W[m]["mean"]
        out[tag + "W%d_cov" % m] = q.W[m]["cov"]
        out[tag + "b%d_mean" % m] = q.b[m]["mean"]
        out[tag + "tau%d" % m] = np.array([q.tau[m]["a"], np.ravel(q.tau[m]["b"])[0]])
        out[tag + "alpha%d_b" % m] = q.alpha[m]["b"]
        out[tag + "X%d_mean" % m] = model.X[m]["mean"]
        out[tag + "X%d_cov" % m] = model.X[m]["cov"]
        out[tag + "xi%d" % m] = q.xi[m]


def run_quiet(fn):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        return fn()


def gen_fa(folder, name, Kc=10, iters=4, rank=5, pretrained_kw=True):
    fav = rh.import_reference(folder)
    X0, lab = fa_inputs(rank=rank)
    model = fav.SSHIBA(Kc, 1)
    v0, v1 = model.struct_data(X0, "reg"), model.struct_data(lab, "mult")
    kw = dict(max_iter=iters, pruning_crit=1e-3, verbose=1, store=False)
    if pretrained_kw:
        kw["pretrained"] = 0
    run_quiet(lambda: model.fit(v0, v1, **kw))
    out = {}
    dump_q(model, out)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "L =", model.L[-3:], "Kc =", model.q_dist.Kc)


def ss_inputs():
    """'reg' view with unobserved entries: the last 128 of 512 rows entirely, plus every 7th entry of column 3."""
    X0, lab = fa_inputs()
    X0 = X0.copy()
    X0[384:, :] = np.nan
    X0[::7, 3] = np.nan
    return X0, lab


def gen_fa_ss(folder, name, pretrained_kw=True, labels_too=False):
    """Semi-supervised 'reg' view (NaN = unobserved, imputed by update_xs every iteration; cA/favae.py:236-257, 775-783);
    labels_too: the multilabel view has unobserved labels as well (update_t, :269-289, 858-861, 1176-1195)."""
    fav = rh.import_reference(folder)
    X0, lab = ss_inputs()
    if labels_too:
        lab = lab.astype(float)
        lab[300:420, :] = np.nan
        lab[::5, 7] = np.nan
    model = fav.SSHIBA(10, 1)
    v0, v1 = model.struct_data(X0, "reg"), model.struct_data(lab, "mult")
    kw = dict(max_iter=4, pruning_crit=1e-3, verbose=1, store=False)
    if pretrained_kw:
        kw["pretrained"] = 0
    run_quiet(lambda: model.fit(v0, v1, **kw))
    out = {}
    dump_q(model, out)
    out["XS0_mean"] = model.q_dist.XS[0]["mean"]
    out["X0_sumlogdet"] = np.array(model.X[0]["sumlogdet"])
    if labels_too:
        out["t1_mean"] = model.t[1]["mean"]
        out["tS1_sumlogdet"] = np.array(model.q_dist.tS[1]["sumlogdet"])
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "L =", model.L[-3:], "Kc =", model.q_dist.Kc)


def gen_predict_fa(folder, name, pretrained_kw=True):
    fav = rh.import_reference(folder)
    X0, lab = fa_inputs()
    model = fav.SSHIBA(10, 1)
    v0, v1 = model.struct_data(X0, "reg"), model.struct_data(lab, "mult")
    kw = dict(max_iter=4, pruning_crit=1e-3, verbose=1, store=False)
    if pretrained_kw:
        kw["pretrained"] = 0
    run_quiet(lambda: model.fit(v0, v1, **kw))
    _, lab_t = fa_inputs(n=100, seed=2)
    p_reg = run_quiet(lambda: model.predict([1], [0], model.struct_data(lab_t, "mult")))
    zm = np.copy(model.Z_mean)
    p_mult = run_quiet(lambda: model.predict([1], [1], model.struct_data(lab_t, "mult")))
    out = dict(Z_mean=zm, reg_mean=p_reg["output_view0"]["mean_x"], reg_var=p_reg["output_view0"]["var_x"],
               mult_mean=p_mult["output_view1"]["mean_x"], mult_var=p_mult["output_view1"]["var_x"],
               Kc=np.array(model.q_dist.Kc))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "Kc", model.q_dist.Kc, "reg_mean", out["reg_mean"].shape, float(np.abs(out["mult_mean"]).mean()))


def gen_vae_step():
    import torch
    vm = rh.import_vae_module("condition_a_VAE", "celeba_deepvanilla_vae")
    torch.manual_seed(0)
    np.random.seed(0)
    vae = run_quiet(lambda: vm.ImgVAE(dimx=20, channels=3, lr=1e-3, h=64, w=64, dataset="g"))
    B = 16
    img, _ = img_inputs(B)
    rs = np.random.RandomState(7)
    prior = torch.from_numpy(rs.randn(B, 20) * 0.5)           # float64, as Z@W^T+b arrives (:209)
    tau = torch.Tensor(np.asarray(np.array([1.7])))           # (:210)
    images = torch.tensor(img)
    out = {}
    vae.train()
    for step in range(3):
        std_P = torch.sqrt(torch.ones_like(prior) * (1 / tau))
        vae.optim.zero_grad()
        torch.manual_seed(100 + step)
        rec, mu_Q, std_Q, z = vae.forward(images)
        rec_err = -vae.gaussian_LL(images, rec)
        kl = vae.kl_div(mu_Q, std_Q, prior, std_P)
        loss = rec_err + 1 * kl
        loss.backward()
        out["s%d_loss" % step] = np.array([loss.item(), rec_err.item(), kl.item()])
        out["s%d_mu" % step] = mu_Q.detach().numpy()
        out["s%d_std" % step] = std_Q.detach().numpy()
        out["s%d_rec" % step] = checksum(rec.detach().numpy())
        for k, p in vae.named_parameters():
            out["s%d_g_%s" % (step, k)] = checksum(p.grad.numpy())
        vae.optim.step()
        for k, p in vae.state_dict().items():
            out["s%d_p_%s" % (step, k)] = checksum(p.numpy())
    np.savez_compressed(os.path.join(OUT, "vae_step.npz"), **out)
    print("vae_step losses", [out["s%d_loss" % s] for s in range(3)])


def gen_vae_epoch():
    import torch
    vm = rh.import_vae_module("condition_a_VAE", "celeba_deepvanilla_vae")
    torch.manual_seed(0)
    np.random.seed(0)
    vae = run_quiet(lambda: vm.ImgVAE(dimx=8, channels=3, lr=1e-3, h=64, w=64, dataset="g"))
    N = 24
    img, _ = img_inputs(N)
    rs = np.random.RandomState(3)
    Z = dict(mean=rs.randn(N, 4), cov=np.eye(4))
    W = dict(mean=rs.randn(8, 4) * 0.3, cov=np.eye(4))
    b = dict(mean=rs.randn(1, 8) * 0.1, cov=np.eye(8))
    tau = np.array([2.5])
    keep = run_quiet(lambda: vae.trainloop(img=img, Z=Z, W=W, b=b, tau=tau, epochs=3, favae=True, beta=1))
    mu, var = run_quiet(lambda: vae.update_x(img=img))
    rec = run_quiet(lambda: vae.reconstruction(mu, var))
    out = dict(keep=np.array(keep), elbo=np.array(vae.elbo_training, dtype=np.float64),
               rec_loss=np.array(vae.reconstruc_during_training, dtype=np.float64),
               kl=np.array(vae.KL_QandP, dtype=np.float64), mu=mu, var=var, rec=checksum(rec))
    for k, p in vae.state_dict().items():
        out["p_" + k] = checksum(p.numpy())
    np.savez_compressed(os.path.join(OUT, "vae_epoch.npz"), **out)
    print("vae_epoch elbo", out["elbo"])


def gen_favae_A():
    fav = rh.import_reference("condition_a_VAE")
    N = 32
    img, lab = img_inputs(N)
    model = fav.SSHIBA(4, 1)
    v0 = model.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g")
    v1 = model.struct_data(lab, "mult")
    run_quiet(lambda: model.fit(v0, v1, max_iter=2, pruning_crit=1e-3, verbose=1, pretrained=0, store=False))
    out = {}
    dump_q(model, out)
    out["elbo_last"] = np.array(model.img_vae[0].elbo_training, dtype=np.float64)
    out["total_elbo0"] = np.array(model.total_elbo[0], dtype=np.float64)
    for k, p in model.img_vae[0].state_dict().items():
        out["p_" + k] = checksum(p.numpy())
    old = os.path.join(OUT, "favae_A.npz")
    if os.path.exists(old):                                  # the fit is deterministic: the committed file must not move
        ref = np.load(old)
        worst = max(float(np.max(np.abs(ref[k] - out[k]))) for k in ref.files)
        print("favae_A re-run vs committed file: max abs diff", worst)
        assert worst == 0.0, "favae_A.npz changed on regeneration"
    else:
        np.savez_compressed(old, **out)
    print("favae_A L =", model.L)
    # ---- predict paths on the trained model (cA/favae.py:1336-1570) ----
    import torch
    img_t, lab_t = img_inputs(12, seed=5)
    pr = {}
    np.random.seed(11); torch.manual_seed(11)
    p = run_quiet(lambda: model.predict([0], [1], model.struct_data(img_t, "img")))
    pr["i2m_Z_mean"], pr["i2m_mean"], pr["i2m_var"] = np.copy(model.Z_mean), p["output_view1"]["mean_x"], p["output_view1"]["var_x"]
    np.random.seed(12); torch.manual_seed(12)
    pred_img, zm, xc = run_quiet(lambda: model.predict([1], [0], model.struct_data(lab_t, "mult")))
    pr["m2i_Z_mean"], pr["m2i_img"] = np.copy(zm), checksum(pred_img.detach().cpu().numpy())
    pr["m2i_img_shape"] = np.array(pred_img.shape)
    np.random.seed(13); torch.manual_seed(13)
    pred_img, zm, xc = run_quiet(lambda: model.predict_sameimage([1], [0], model.struct_data(lab_t, "mult")))
    pr["same_Z_mean"], pr["same_img"], pr["same_xcart"] = np.copy(zm), checksum(pred_img), checksum(xc.numpy())
    np.savez_compressed(os.path.join(OUT, "favae_A_predict.npz"), **pr)
    print("favae_A_predict: img->mult mean", float(pr["i2m_mean"].mean()), "mult->img", pr["m2i_img"][:2], pr["same_img"][:2])


def gen_favae_C():
    fav = rh.import_reference("transfer_learning")
    N = 16
    img, lab = img_inputs(N)
    model = fav.SSHIBA(4, 1)
    X0, _ = fa_inputs(n=N, seed=3)            # stands in for the pretrained-VAE latents of TL/trainer_FAVAE.py:24-31
    v0 = model.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g")
    v1 = model.struct_data(X0, "reg")
    v2 = model.struct_data(lab[:, :5], "mult")
    run_quiet(lambda: model.fit(v0, v1, v2, max_iter=2, pruning_crit=1e-3, verbose=1, store=False))   # view order of :96
    out = {}
    dump_q(model, out)
    for i in range(len(model.total_elbo)):
        out["total_elbo%d" % i] = np.array(model.total_elbo[i], dtype=np.float64)
        out["kl%d" % i] = np.array(model.kl_q_p[i], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "favae_C.npz"), **out)
    print("favae_C L =", model.L, "elbo lists", len(model.total_elbo))


def gen_vae_beta_epoch():
    import torch
    vm = rh.import_vae_module("transfer_learning", "betaVAE")
    torch.manual_seed(0)
    np.random.seed(0)
    vae = run_quiet(lambda: vm.ImgVAE(dimx=8, channels=3, lr=1e-3, h=64, w=64, dataset="g"))
    N = 24
    img, _ = img_inputs(N)
    keep = run_quiet(lambda: vae.trainloop(img=img, epochs=3, beta=4))
    mu, var = run_quiet(lambda: vae.update_x(img=img))
    out = dict(keep=np.array(keep), elbo=np.array(vae.elbo_training, dtype=np.float64),
               rec_loss=np.array(vae.reconstruc_during_training, dtype=np.float64),
               kl=np.array(vae.KL_QandP, dtype=np.float64), mu=mu, var=var)
    np.savez_compressed(os.path.join(OUT, "vae_beta_epoch.npz"), **out)
    print("vae_beta_epoch elbo", out["elbo"], "kl", out["kl"])


def gen_favae_B():
    import torch
    fav = rh.import_reference("domain_adaptation")
    N = 16
    img0, lab = img_inputs(N)
    img1, _ = img_inputs(N, seed=4, zt=np.random.RandomState(1).randn(N, 5))     # second domain, same latent factors
    lab = lab[:, :3]
    model = fav.SSHIBA(4, 1)
    v0 = model.struct_data(img0, "img", latent_dim=8, lr=1e-3, dataset="g0")
    v1 = model.struct_data(img1, "img", latent_dim=8, lr=1e-3, dataset="g1")
    v2 = model.struct_data(lab, "mult")
    run_quiet(lambda: model.fit(v0, v1, v2, max_iter=2, pruning_crit=1e-3, verbose=1, store=False))
    out = {}
    dump_q(model, out)
    for i in range(len(model.total_elbo)):
        out["total_elbo%d" % i] = np.array(model.total_elbo[i], dtype=np.float64)
    for m in (0, 1):
        for k, p in model.img_vae[m].state_dict().items():
            out["p%d_%s" % (m, k)] = checksum(p.numpy())
    img_t, _ = img_inputs(N, seed=5)          # DA/celeba2cartoon_vae.update_x encodes self.img whatever it is handed (:240-258),
    np.random.seed(21); torch.manual_seed(21)   # so predict only runs with n_pred == N
    pred_img, X_sample, Z_in, xcart = run_quiet(lambda: model.predict([0], [1], model.struct_data(img_t, "img")))
    out["pred_img"], out["pred_X_sample"], out["pred_Z_mean"] = checksum(pred_img), X_sample, Z_in["mean"]
    out["pred_Z_cov"], out["pred_xcart"] = Z_in["cov"], checksum(xcart.numpy())
    np.savez_compressed(os.path.join(OUT, "favae_B.npz"), **out)
    print("favae_B L =", model.L, "elbo lists", len(model.total_elbo), "pred", out["pred_img"][:2])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["fa", "vae_step", "vae_epoch", "favae_A"]
    if "fa" in which:
        gen_fa("condition_a_VAE", "fa_A.npz")
        gen_fa("domain_adaptation", "fa_B.npz", pretrained_kw=False)
        gen_fa("condition_a_VAE", "fa_A_prune.npz", Kc=30, iters=40, rank=3)
    if "fa_ss" in which:
        gen_fa_ss("condition_a_VAE", "fa_A_ss.npz")
        gen_fa_ss("domain_adaptation", "fa_B_ss.npz", pretrained_kw=False)
        gen_fa_ss("condition_a_VAE", "fa_A_ss2.npz", labels_too=True)
    if "predict_fa" in which:
        gen_predict_fa("condition_a_VAE", "predict_fa_A.npz")
        gen_predict_fa("domain_adaptation", "predict_fa_B.npz", pretrained_kw=False)
    if "vae_step" in which:
        gen_vae_step()
    if "vae_epoch" in which:
        gen_vae_epoch()
    if "favae_C" in which:
        gen_favae_C()
    if "vae_beta" in which:
        gen_vae_beta_epoch()
    if "favae_B" in which:
        gen_favae_B()
    if "favae_A" in which:
        gen_favae_A()
