"""Grouped-batch mode of the conv-VAE step (favae_vae_train_fwdbwd / favae_vae_train_epoch with a batch above 64): G BatchNorm
groups of 64 images in one set of launches, the loss the mean over all G x 64 images, one optimiser step.

The reference has no such launch (its DataLoader is batch_size=64, cA/celeba_deepvanilla_vae.py:218); the mode is DEFINED as the
average of G reference mini-batch steps taken from the same parameters (= the G-rank data-parallel step), so the oracle here is the
CPU oracle of the reference step (oracle/vae_torch.py, pinned by tests/test_oracle_golden.py) run once per group and averaged by
hand.  Tolerances as in tests/test_gpu_vae_step.py: fp32 1e-4 / bf16 1e-2 on loss terms and latent moments."""
import numpy as np
import pytest
import torch

from conftest import img_inputs
from oracle import vae_torch as vt
from test_gpu_vae_step import NOISE_ONLY, _cos, _rel, _rel_p99

pytestmark = pytest.mark.gpu


def _oracle_groups(P, Bf, images, prior, tau, eps, sizes):
    """One reference step per group from the same state: mean loss terms, mean gradient (left in P[k].grad), mean of the groups'
    updated running statistics, concatenated latent moments."""
    for p in P.values():
        p.grad = None
    tot = float(sum(sizes))
    rec_m = kl_m = 0.0
    run = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in Bf.items() if "num_batches" not in k}
    mus, stds = [], []
    o = 0
    for n in sizes:
        Bg = {k: v.clone() for k, v in Bf.items()}
        loss, rec_err, kl, mu, std, zz, rec = vt.loss_terms(P, Bg, images[o:o + n], prior[o:o + n], tau, 0.01, 1, eps=eps[o:o + n])
        (loss * (n / tot)).backward()           # mean over all images of the step = size-weighted mean of the group means
        rec_m += rec_err.item() * n / tot
        kl_m += kl.item() * n / tot
        for k in run:
            run[k] += Bg[k].double() / len(sizes)
        mus.append(mu.detach())
        stds.append(std.detach())
        o += n
    return rec_m, kl_m, run, torch.cat(mus), torch.cat(stds)


@pytest.mark.parametrize("dtype,tol,tol_bulk,min_cos", [("f32", 1e-4, 1e-2, 0.9999), ("bf16", 1e-2, 0.5, 0.97)])
@pytest.mark.parametrize("G", [2, 3])
def test_grouped_step_is_the_average_of_its_groups(dtype, tol, tol_bulk, min_cos, G):
    from favae_hip.engine import VaeEngine
    z, Bn = 20, 64 * G
    torch.manual_seed(0)
    np.random.seed(0)
    P, Bf = vt.init_state(z, 3)
    for k in Bf:                                   # non-trivial running statistics, so that their update is visible
        if k.endswith("running_var"):
            Bf[k].uniform_(0.5, 1.5)
        elif k.endswith("running_mean"):
            Bf[k].normal_(0, 0.1)
    names = list(P.keys())
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.clone() for k, v in Bf.items()})
    img, _ = img_inputs(Bn)
    images = torch.tensor(img)
    prior = torch.from_numpy(np.random.RandomState(7).randn(Bn, z) * 0.5)
    tau = torch.Tensor(np.asarray(np.array([1.7])))
    eps = torch.randn(Bn, z, generator=torch.Generator().manual_seed(11))
    rec_m, kl_m, run, mu, std = _oracle_groups(P, Bf, images, prior, tau, eps, [64] * G)

    eng = VaeEngine(z, 3, batch_max=Bn, dtype=dtype)
    eng.load_state_dict(sd)
    data = eng.upload_images(img)
    idx = torch.arange(Bn, dtype=torch.int32, device="cuda")
    m = eng.train_fwdbwd(data, idx, Bn, eps.cuda(), prior.cuda(), 1.7, 1.0, 0.01).cpu().numpy()
    assert abs(m[0] - rec_m) <= tol * abs(rec_m), (m, rec_m)
    assert abs(m[1] - kl_m) <= tol * abs(kl_m) + 1e-6, (m, kl_m)
    fc = eng.debug(0, 5, (Bn, 2 * z), torch.float32).cpu().numpy()
    assert _rel(fc[:, :z], mu.numpy()) <= tol, _rel(fc[:, :z], mu.numpy())
    assert _rel(fc[:, z:] + 1e-6, 2 * np.log(std.numpy())) <= tol
    g = eng.unpack(eng.G)
    for k in names:
        if k in NOISE_ONLY:
            continue
        gb, gr = g[k].numpy(), P[k].grad.numpy()
        assert _rel_p99(gb, gr) <= tol_bulk, "gradient %s bulk rel err %.3e" % (k, _rel_p99(gb, gr))
        assert _cos(gb, gr) >= min_cos, "gradient %s cosine %.6f" % (k, _cos(gb, gr))
    got = eng.state_dict()
    for k in run:
        assert _rel(got[k].numpy(), run[k].numpy()) <= tol, (k, _rel(got[k].numpy(), run[k].numpy()))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_grouped_epoch_with_a_ragged_tail_follows_the_averaged_steps(dtype):
    """An epoch of 356 images at 128 per step: steps of 128, 128, then the tail of 100 as its whole group (64) and the rest (36, an
    ordinary step).  Against the oracle stepping Adam on the hand-averaged group gradients.  fp32; the first step to 1e-4, later
    steps to 5e-3 (measured up to 2.0e-3) (Adam's first steps move every entry by ~lr whatever its gradient, so entries with rounding-noise gradients
    differ between any two runs: tools/dbg_epoch_vs_steps.py).  bf16 (conv outputs of the decoder and activation gradients stored as
    bf16 in this mode; the ordinary tail step runs on the same storage): loss terms to 1e-2 / 3e-2."""
    from favae_hip.engine import VaeEngine
    z, n, batch = 20, 356, 128
    assert VaeEngine.epoch_steps(n, batch) == 4 and VaeEngine.epoch_steps(256, batch) == 2 and VaeEngine.epoch_steps(300, batch) == 3
    assert VaeEngine.epoch_steps(n, 64) == 6 and VaeEngine.epoch_steps(64 * 5, 64 * 5) == 1
    torch.manual_seed(2)
    np.random.seed(2)
    P, Bf = vt.init_state(z, 3)
    names = list(P.keys())
    sd = {k: v.detach().clone() for k, v in P.items()}
    sd.update({k: v.clone() for k, v in Bf.items()})
    img, _ = img_inputs(n)
    prior = torch.from_numpy(np.random.RandomState(7).randn(n, z) * 0.5)
    tau = torch.Tensor(np.asarray(np.array([1.7])))
    eps = torch.randn(n, z, generator=torch.Generator().manual_seed(5))
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(6))

    eng = VaeEngine(z, 3, batch_max=batch, dtype=dtype)
    eng.load_state_dict(sd)
    data = eng.upload_images(img)
    ms = torch.zeros(4, 2, dtype=torch.float64, device="cuda")
    eng.train_epoch(data, perm.to(torch.int32).cuda(), eps.cuda(), prior.cuda(), 1.7, 1.0, 0.01, 1e-4, batch, ms)
    torch.cuda.synchronize()
    assert eng.step == 4
    got_m = ms.cpu().numpy()

    opt = torch.optim.Adam([P[k] for k in names], 1e-4)     # (1e-3 from a fresh initialisation blows the KL term up to 1e13 by step 3)
    images = torch.tensor(img)[perm]
    prior_v = prior[perm]                           # the executor reads prior rows by sample index
    o = 0
    for s, sizes in enumerate([[64, 64], [64, 64], [64], [36]]):
        B = sum(sizes)
        rec_m, kl_m, run, mu, std = _oracle_groups(P, Bf, images[o:o + B], prior_v[o:o + B], tau, eps[o:o + B], sizes)
        opt.step()
        for k in run:
            Bf[k].copy_(run[k].to(Bf[k].dtype))
        tol = (1e-4 if s == 0 else 5e-3) if dtype == "f32" else (1e-2 if s == 0 else 3e-2)
        assert abs(got_m[s, 0] - rec_m) <= tol * abs(rec_m), (s, got_m[s], rec_m)
        assert abs(got_m[s, 1] - kl_m) <= tol * abs(kl_m) + 1e-6, (s, got_m[s], kl_m)
        o += B
    got = eng.state_dict()
    if dtype != "f32":
        return
    for k in names:
        if k in NOISE_ONLY:
            continue
        d = np.abs(got[k].numpy() - P[k].detach().numpy())
        # (medians measured over 8 runs: 3e-6 .. 7e-6 for every tensor; 1.1e-5 seen once for the 64 entries of enc.0's BatchNorm weight)
        assert np.median(d) <= 2.5e-5 and d.max() <= 4 * 2.1e-4, (k, d.max(), np.median(d))
    for k in Bf:
        if "num_batches" in k:
            continue
        # The running mean follows the bias of the convolution in front of the BatchNorm.  That bias has a zero gradient, which the
        # grouped step keeps exactly (the bias stays put), while autograd's rounding noise, normalised by Adam, moves the oracle's
        # bias by ~lr per step (4e-4 after this epoch, 1/3 of it in the running mean, against running means of ~1e-2).
        # (after four Adam steps; the update rule itself is held to 1e-4 in the one-step test above.  Measured 3.2e-3 for enc.4's
        # running variance, 1.3e-2 for dec.0's running mean)
        tol = 3e-2 if k.endswith("running_mean") else 1e-2
        assert _rel(got[k].numpy(), Bf[k].numpy()) <= tol, (k, _rel(got[k].numpy(), Bf[k].numpy()))


def test_a_batch_above_64_must_be_whole_groups():
    from favae_hip.engine import VaeEngine
    from favae_hip.lib import FavaeHipError
    eng = VaeEngine(20, 3, batch_max=128, dtype="f32")
    img, _ = img_inputs(100)
    data = eng.upload_images(img)
    idx = torch.arange(100, dtype=torch.int32, device="cuda")
    z = torch.zeros(100, 20, device="cuda")
    with pytest.raises(FavaeHipError, match="groups of 64"):
        eng.train_fwdbwd(data, idx, 100, z, None, 1.0, 1.0, 0.01)


def test_imgvae_groups_option_runs_the_grouped_epoch():
    """ImgVAE.GROUPS = 2 through the drop-in class (trainloop with no FA prior, as the pretraining call of cA/favae.py:690-700 makes
    it): same visiting order and noise as the batch-64 trainloop, steps of 128 images.  Against the executor's epoch entry point
    called directly with the replayed random draws from the same initial state (the plumbing: step count, metric averaging,
    executor size), to 1e-4 of the epoch ELBO (same kernels; atomics reorder sums and Adam amplifies that over the three steps)."""
    from favae_hip import imgvae
    from favae_hip.engine import VaeEngine

    class Grouped(imgvae.ImgVAE):
        GROUPS = 2

    n, z = 300, 20
    img, _ = img_inputs(n)
    torch.manual_seed(4)
    np.random.seed(4)
    vae = Grouped(dimx=z, channels=3, lr=1e-4, dataset="grouped", precision="f32")
    sd0 = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    state = torch.get_rng_state()
    vae.trainloop(img, epochs=1, favae=False)
    assert vae._eng.batch_max == 128 and vae._eng.step == 3            # 128 + 128 + 44
    torch.set_rng_state(state)                                            # replay the draws of the epoch (imgvae.py trainloop)
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    perm = torch.randperm(n, generator=g)
    eps = torch.cat([torch.randn(min(64, n - s), z) for s in range(0, n, 64)], 0)
    eng = VaeEngine(z, 3, batch_max=128, dtype="f32")
    eng.load_state_dict(sd0)
    data = eng.upload_images(img)
    ms = torch.zeros(3, 2, dtype=torch.float64, device="cuda")
    eng.train_epoch(data, perm.to(torch.int32).cuda(), eps.cuda(), None, 1.0, 1.0, 0.01, 1e-4, 128, ms)
    m = ms.cpu().numpy()
    want = -(m[:, 0] + m[:, 1]).sum() / 3
    assert abs(vae.elbo_training[0] - want) <= 1e-4 * abs(want), (vae.elbo_training[0], want)   # (measured 2.8e-6)
