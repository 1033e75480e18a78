"""The drop-in boundary exercised the way the reference's trainers call it, against the oracle doing the same:

* cA/vanillavae_trainer.py:17-40 -- a torch TENSOR handed to the stand-alone `ImgVAE.trainloop`, the model pickled, un-pickled,
  `img_vae.img[0]`, `img_vae.eval()`, `encoder(x.to(device))`, `decoder(z)` on the un-pickled model;
* cA/condition_a_pretrainedvae.py:83-102 -- `fit(pretrained=1)`, the un-pickled VAE injected as `img_vae[0]`, `update_x()` with no
  argument, `X[0]` assigned by hand, `fit_iterate(...)`;
* the reference's `./checkpoints/*.pt` file written after a real training loop on the GPU and read by the oracle's module;
* the bf16 conv path through `SSHIBA.fit` at the tolerance north_star states for it (1e-2 on ELBO, latent means, W)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import img_inputs
from oracle import fa_numpy as fo
from oracle import vae_torch as vt
from test_gpu_sshiba import _close, load_variant

pytestmark = pytest.mark.gpu


def _adam_state_to_oracle(vae, ref):
    """Give the oracle's module the HIP model's weights, BN buffers and Adam moments (like-for-like continuation)."""
    sd = {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}
    ref.load_state_dict(sd)
    names = dict(vae.named_parameters())
    for k in ref.names:
        st = vae.optim.state[names[k]]
        ref.optim.state[ref.P[k]] = {"step": st["step"].clone(), "exp_avg": st["exp_avg"].detach().cpu().clone(),
                                     "exp_avg_sq": st["exp_avg_sq"].detach().cpu().clone()}


def test_pretrained_vae_injection_flow_matches_oracle(tmp_path):
    """vanillavae_trainer.py -> pickle -> condition_a_pretrainedvae.py, shortened (N = 48, latent 100 as in the trainers, one
    epoch each), with the reference's calls verbatim; the oracle takes over the pre-trained state so that every compared
    quantity is like for like (fp32 path: 1e-4 / 1e-5 for the first epoch from an identical state)."""
    FAVAE = load_variant("condition_a_VAE")
    import celeba_deepvanilla_vae as vae
    N, z = 48, 100
    img, lab = img_inputs(N)
    lab = lab[:, :3]
    celeba_train = torch.from_numpy(img)                       # the stand-alone trainers hand over a tensor (:18-19)
    # ---- vanillavae_trainer.py:23-34 ----
    torch.manual_seed(0)
    np.random.seed(0)
    img_vae = vae.ImgVAE(dimx=z, channels=3, lr=1e-3, h=celeba_train.shape[2], w=celeba_train.shape[3], dataset="celeba")
    img_vae.trainloop(img=celeba_train, Z=None, W=None, b=None, tau=None, epochs=1, wandb=None, favae=False)
    blob = pickle.dumps({"model": img_vae}, protocol=pickle.HIGHEST_PROTOCOL)
    torch.manual_seed(0)
    np.random.seed(0)
    ref_vae = vt.OracleImgVAE(dimx=z, channels=3, lr=1e-3, dataset="celeba")
    ref_vae.trainloop(img=img, Z=None, W=None, b=None, tau=None, epochs=1, favae=False)
    _close(img_vae.elbo_training[0], ref_vae.elbo_training[0], 1e-5)
    # ---- the checkpoint file of restore_weights, written after a real training loop, read by the oracle's module ----
    path = img_vae.save_checkpoint(str(tmp_path / "checkpoints" / "celeba_celeba_vanilla_1.pt"))
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "elbo", "state_dict", "optimizer"} and ck["epoch"] == 1
    _close(ck["elbo"], img_vae.elbo_training[0], 1e-12)
    chk = vt.OracleImgVAE(dimx=z, channels=3, lr=1e-3, dataset="celeba")
    chk.load_state_dict(ck["state_dict"])
    chk.img = torch.tensor(img)
    assert len(ck["optimizer"]["state"]) == len(chk.names)
    for i, k in enumerate(chk.names):                          # Adam moments in the reference's parameter order
        assert tuple(ck["optimizer"]["state"][i]["exp_avg"].shape) == tuple(chk.P[k].shape), k
    # ---- vanillavae_trainer.py:36-70 on the un-pickled model ----
    back = pickle.loads(blob)["model"]
    example = back.img[0, :, :, :].permute(1, 2, 0).detach().numpy()
    assert example.shape == (64, 64, 3) and np.array_equal(example, img[0].transpose(1, 2, 0))
    back.eval()
    x = back.img[0, :, :, :].unsqueeze(0).to(back.device)
    mu, std = back.encoder(x)
    eps = torch.normal(0, 5, size=std.size(), generator=torch.Generator().manual_seed(3)).to(back.device)
    zs = eps.mul(std).add_(mu)
    x_hat = back.decoder(zs)
    with torch.no_grad():
        mu_r, std_r = vt.encoder(torch.tensor(img[:1]), chk.P, chk.B, False)
        xh_r = vt.decoder(zs.cpu(), chk.P, chk.B, False)
    _close(mu.cpu(), mu_r, 0, 1e-4 * float(mu_r.abs().max()))
    _close(std.cpu(), std_r, 1e-4, 1e-6)
    _close(x_hat.cpu(), xh_r, 0, 1e-4)
    # ---- condition_a_pretrainedvae.py:76-102 ----
    outs = []
    pre = vt.OracleImgVAE(dimx=z, channels=3, lr=1e-3, dataset="celeba")      # the oracle's "un-pickled" model: built before the
    _adam_state_to_oracle(img_vae, pre)                                        # seeds are set (its construction draws)
    pre.img = torch.tensor(img)
    for side in ("hip", "oracle"):
        np.random.seed(0)
        torch.manual_seed(0)
        if side == "hip":
            FAVAE_model = FAVAE.SSHIBA(6, 1, latentspace_lr=1)
            celeba_in = FAVAE_model.struct_data(img, "img", latent_dim=z, lr=1e-3, dataset="celeba")
            attributes = FAVAE_model.struct_data(lab, "mult")
            FAVAE_model.fit(celeba_in, attributes, pretrained=1)
            modelpre = pickle.loads(blob)
        else:
            FAVAE_model = fo.SSHIBAOracle(6, 1, variant="A", vae_factory=lambda **kw: vt.OracleImgVAE(**kw))
            FAVAE_model.fit(fo.struct_data(img, "img", latent_dim=z, lr=1e-3, dataset="celeba"), fo.struct_data(lab, "mult"), iterate=False)
            modelpre = {"model": pre}
        assert FAVAE_model.L == []
        FAVAE_model.img_vae[0] = modelpre["model"]
        FAVAE_model.X[0]["mean"], FAVAE_model.X[0]["cov"] = modelpre["model"].update_x()
        FAVAE_model.X[0]["prodT"] = np.dot(FAVAE_model.X[0]["mean"].T, FAVAE_model.X[0]["mean"]) + np.diag(np.sum(FAVAE_model.X[0]["cov"], axis=0))
        x_pre = (np.array(FAVAE_model.X[0]["mean"]), np.array(FAVAE_model.X[0]["cov"]))
        FAVAE_model.epochs_override = [1]
        if side == "hip":
            FAVAE_model.fit_iterate(max_iter=1, pruning_crit=1e-3, verbose=1, store=False)
        else:
            FAVAE_model.fit_iterate(max_iter=1, pruning_crit=1e-3)
        outs.append((FAVAE_model, x_pre))
    (model, xh), (ref, xr) = outs
    _close(xh[0], xr[0], 0, 1e-4 * np.abs(xr[0]).max())       # update_x() of the un-pickled model: latent means
    _close(xh[1], xr[1], 1e-4, 1e-7)
    _close(model.total_elbo[0][0], ref.total_elbo[0][0], 1e-5)  # first epoch of the coupled training: identical state going in
    _close(model.kl_q_p[0][0], ref.kl_q_p[0][0], 1e-4)
    assert len(model.L) == 1
    _close(model.L[0], ref.L[0], 2e-3)
    _close(model.q_dist.Z["mean"], ref.q_dist.Z["mean"], 0, 1e-4 * np.abs(ref.q_dist.Z["mean"]).max())   # from the injected X (fp64 algebra)
    store = pickle.dumps({"model": model}, protocol=pickle.HIGHEST_PROTOCOL)   # :107-109
    assert pickle.loads(store)["model"].L == model.L


def test_image_view_as_a_torch_tensor_equals_the_numpy_view():
    """struct_data / trainloop / update_x take the images as a torch tensor (what the trainers hold before `.numpy()`,
    cA/favae_trainer.py:21-28; the stand-alone trainers pass it on as it is) with the same result as the NumPy array."""
    FAVAE = load_variant("condition_a_VAE")
    N = 32
    img, lab = img_inputs(N)
    res = []
    for view in (img, torch.from_numpy(img)):
        np.random.seed(0)
        torch.manual_seed(0)
        model = FAVAE.SSHIBA(4, 1)
        model.epochs_override = [1]
        model.fit(model.struct_data(view, "img", latent_dim=8, lr=1e-3, dataset="g"), model.struct_data(lab, "mult"), max_iter=1,
                  pruning_crit=1e-3, verbose=0, store=False, pretrained=0)
        mu, var = model.img_vae[0].update_x(view)
        res.append((model.total_elbo[0][0], model.L[0], mu, var, np.array(model.X[0]["mean"])))
    _close(res[1][0], res[0][0], 1e-6)                         # the epoch ran on identical batches
    _close(res[1][1], res[0][1], 1e-3)
    for a, b in zip(res[1][2:], res[0][2:]):
        # two runs of the SAME path: the order of split-K atomics decides the sign of near-zero gradient entries, and Adam's first
        # step moves every weight by +-lr whatever the magnitude (measured 2.3e-2 of scale between two NumPy-view runs)
        _close(a, b, 0, 0.1 * np.abs(b).max())


def _take_over_everything(model, ref, img):
    """The oracle continues from the HIP model's state: VAE weights / BN buffers / Adam moments, the posterior, and X of the
    image view recomputed by the ORACLE's fp32 encoder on those weights (so the next iteration differs by the conv path only)."""
    q, rq = model.q_dist, ref.q_dist
    rq.Kc = q.Kc
    for k in ("mean", "cov", "prodT"):
        rq.Z[k] = np.array(q.Z[k])
    for m in range(model.m):
        for k in ("mean", "cov", "prodT"):
            rq.W[m][k] = np.array(q.W[m][k])
            rq.b[m][k] = np.array(q.b[m][k])
        rq.tau[m]["a"], rq.tau[m]["b"] = q.tau[m]["a"], np.array(q.tau[m]["b"])
        rq.alpha[m]["a"], rq.alpha[m]["b"] = q.alpha[m]["a"], np.array(q.alpha[m]["b"])
        rq.xi[m] = q.xi[m].cpu().numpy().copy()
        if model.img_vae[m] is not None:
            model.img_vae[m].sync()
            _adam_state_to_oracle(model.img_vae[m], ref.img_vae[m])
            mu, var = ref.img_vae[m].update_x(img)
            ref.X[m]["mean"], ref.X[m]["cov"] = mu, var
            ref.X[m]["prodT"] = mu.T @ mu + np.diag(np.sum(var, axis=0))
        else:
            for k in ("mean", "cov", "prodT"):
                ref.X[m][k] = np.array(model.X[m][k])
    ref.keep_training = dict(model.keep_training)


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("bf16", 1e-2)])
def test_full_variant_A_latents_and_W_at_the_stated_tolerance(precision, tol, monkeypatch):
    """north_star: rel <= 1e-4 (fp32) / <= 1e-2 (bf16 conv path) on the ELBO, the latent means and W.  One image view (z = 8) +
    the multilabel view, N = 64, Kc = 4.  Iteration 1 trains one epoch from the shared initialisation (ELBO like for like).
    Then the oracle takes over the HIP model's state and both run iteration 2 -- W and Z updates from X of the image view,
    i.e. from the latent means of update_x -- where the only difference is the conv path: the HIP side holds the latent means
    of its own (bf16 / fp32) eval encoder, the oracle those of torch fp32 on the same weights."""
    monkeypatch.setenv("FAVAE_CONV_DTYPE", precision)
    fav = load_variant("condition_a_VAE")
    N = 64
    img, lab = img_inputs(N)
    np.random.seed(0)
    torch.manual_seed(0)
    ref = fo.SSHIBAOracle(4, 1, variant="A", vae_factory=lambda **kw: vt.OracleImgVAE(**kw))
    ref.epochs_override = [1]
    ref.fit(fo.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g"), fo.struct_data(lab, "mult"), max_iter=1, pruning_crit=1e-3)
    np.random.seed(0)
    torch.manual_seed(0)
    model = fav.SSHIBA(4, 1)
    model.epochs_override = [1]
    model.fit(model.struct_data(img, "img", latent_dim=8, lr=1e-3, dataset="g"), model.struct_data(lab, "mult"), max_iter=1,
              pruning_crit=1e-3, verbose=0, store=False, pretrained=0)
    assert model.img_vae[0].precision == precision
    _close(model.total_elbo[0][0], ref.total_elbo[0][0], tol)                   # ELBO of the first epoch
    _close(model.recon_loss[0][0], ref.recon_loss[0][0], tol)
    _close(model.kl_q_p[0][0], ref.kl_q_p[0][0], tol)
    _take_over_everything(model, ref, img)
    xm, xr = np.array(model.X[0]["mean"]), ref.X[0]["mean"]
    _close(xm, xr, 0, tol * np.abs(xr).max())                                   # latent means on the same weights
    _close(model.X[0]["cov"], ref.X[0]["cov"], tol, tol * np.abs(ref.X[0]["cov"]).max())
    assert not model.keep_training[0]          # one epoch: |1 - mean/last| = 0 (:287-289), so iteration 2 is FA algebra on X
    for s in (model, ref):
        np.random.seed(5)
        torch.manual_seed(5)
        s.update(iter_count=1)
        s.depruning(1e-15)
        s.L.append(s.update_bound())
        s.pruning(1e-3)
    assert model.q_dist.Kc == ref.q_dist.Kc
    _close(model.L[1], ref.L[1], tol)                                           # the bound
    for m in range(2):
        w = ref.q_dist.W[m]["mean"]
        _close(model.q_dist.W[m]["mean"], w, 0, tol * np.abs(w).max())          # W
        _close(model.q_dist.tau_mean(m), ref.q_dist.tau_mean(m), tol)
    zr = ref.q_dist.Z["mean"]
    _close(model.q_dist.Z["mean"], zr, 0, tol * np.abs(zr).max())               # Z


def test_unpickled_model_predicts_and_resumes():
    """The result scripts un-pickle a fitted model and call predict (DA/results_celeba2cartoon.py:25-34); the device state a
    pickle drops is rebuilt on first use, and fit_iterate on the un-pickled model goes on exactly where the original does."""
    from conftest import fa_inputs
    fav = load_variant("condition_a_VAE")
    np.random.seed(0)
    X0, lab = fa_inputs()
    model = fav.SSHIBA(10, 1)
    model.fit(model.struct_data(X0, "reg"), model.struct_data(lab, "mult"), max_iter=2, pruning_crit=1e-3, verbose=0, store=False,
              pretrained=0)
    back = pickle.loads(pickle.dumps({"model": model}, protocol=pickle.HIGHEST_PROTOCOL))["model"]
    assert "dev" not in back.__dict__
    _, lab_t = fa_inputs(n=100, seed=2)
    pa = model.predict([1], [0], model.struct_data(lab_t, "mult"))
    pb = back.predict([1], [0], back.struct_data(lab_t, "mult"))
    _close(pb["output_view0"]["mean_x"], pa["output_view0"]["mean_x"], 1e-12, 1e-14)
    model.fit_iterate(max_iter=2, pruning_crit=1e-3, verbose=0, store=False)
    back.fit_iterate(max_iter=2, pruning_crit=1e-3, verbose=0, store=False)
    assert len(back.L) == 4
    _close(back.L, model.L, 1e-10)
    _close(back.q_dist.W[1]["mean"], model.q_dist.W[1]["mean"], 1e-9, 1e-12)
