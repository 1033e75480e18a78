"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of include/favae_hip.h, the drop-in
modules expose the reference's surface, the host mirror draws the same initial values as the oracle, and the
multi-process sharding glue (gloo, world_size 2) reproduces un-sharded statistics."""
import inspect
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT


def _load(folder, name):
    import importlib
    for k in ("favae", "celeba_deepvanilla_vae", "celeba_bettervanilla_vae", "celeba2cartoon_vae", "betaVAE"):
        sys.modules.pop(k, None)
    sys.path[:] = [p for p in sys.path if not (p.startswith(PKG + os.sep) and os.path.basename(p) != "favae_hip")]
    sys.path.insert(0, os.path.join(PKG, folder))
    return importlib.import_module(name)


def test_library_exports_every_declared_symbol():
    from favae_hip import lib
    raw = lib.raw()
    text = open(os.path.join(ROOT, "include", "favae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(favae_[a-z0-9_]+)\s*\(", text)))
    assert len(names) > 40
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert raw.favae_version() >= 1
    assert isinstance(lib.last_error(), str)


def test_argument_errors_are_reported_without_a_gpu():
    from favae_hip import lib
    import ctypes
    rc = lib.raw().favae_adam_step(None, None, None, None, ctypes.c_longlong(3), ctypes.c_float(1e-3), ctypes.c_float(0.9),
                                   ctypes.c_float(0.999), ctypes.c_float(1e-8), ctypes.c_longlong(1), None)
    assert rc < 0 and "favae_adam_step" in lib.last_error()
    with pytest.raises(lib.FavaeHipError):
        lib.check(rc)


EXPECTED_TRAINLOOP = {
    ("condition_a_VAE", "celeba_deepvanilla_vae"): ["img", "Z", "W", "b", "tau", "epochs", "wandb", "favae", "beta"],
    ("condition_a_VAE", "celeba_bettervanilla_vae"): ["img", "Z", "W", "b", "tau", "epochs", "wandb", "beta", "favae"],
    ("transfer_learning", "celeba_bettervanilla_vae"): ["img", "Z", "W", "b", "tau", "epochs", "wandb", "beta", "favae"],
    ("domain_adaptation", "celeba2cartoon_vae"): ["img", "Z", "W", "b", "tau", "epochs", "beta", "wandb"],
    ("transfer_learning", "betaVAE"): ["img", "Z", "W", "b", "tau", "epochs", "wandb", "beta", "favae"],
}


@pytest.mark.parametrize("folder,name", list(EXPECTED_TRAINLOOP))
def test_plugin_modules_keep_the_reference_surface(folder, name):
    """Signatures as in the reference plugins (celeba_deepvanilla_vae.py:109-119,173-184 and siblings)."""
    mod = _load(folder, name)
    assert list(inspect.signature(mod.ImgVAE.trainloop).parameters)[1:] == EXPECTED_TRAINLOOP[(folder, name)]
    assert list(inspect.signature(mod.ImgVAE.__init__).parameters)[1:9] == ["dimx", "channels", "var_x", "lr", "h", "w", "dataset",
                                                                          "latentspace_lr"]
    for meth in ("update_x", "reconstruction", "predict", "encoder", "decoder", "forward", "kl_div", "gaussian_LL",
                 "restore_weights", "reparameterize"):
        assert callable(getattr(mod.ImgVAE, meth))
    assert issubclass(mod.ImgVAE, mod.VAE) and issubclass(mod.VAE, torch.nn.Module)


@pytest.mark.parametrize("folder,variant,plugin", [("condition_a_VAE", "A", "celeba_deepvanilla_vae"),
                                                   ("domain_adaptation", "B", "celeba2cartoon_vae"),
                                                   ("transfer_learning", "C", "betaVAE")])
def test_favae_modules_keep_the_reference_surface(folder, variant, plugin):
    fav = _load(folder, "favae")
    assert fav.SSHIBA.VARIANT == variant and fav.img_vae.__name__ == plugin
    assert list(inspect.signature(fav.SSHIBA.__init__).parameters)[1:] == [
        "Kc", "prune", "fs", "SS_sep", "hyper", "X_init", "b_init", "Z_init", "W_init", "alpha_init", "tau_init", "gamma_init",
        "area_mask", "latentspace_lr"]
    assert list(inspect.signature(fav.SSHIBA.struct_data).parameters)[1:] == [
        "X", "method", "sparse", "V", "kernel", "sig", "sparse_fs", "center", "mask", "deg", "latent_dim", "lr", "epochs", "dataset"]
    m = fav.SSHIBA(7, 1, latentspace_lr=0.5)
    v = m.struct_data(np.zeros((3, 2)), "reg")
    assert v["method"] == "reg" and v["SV"] is None and v["kernel"] is None
    for meth in ("fit", "fit_iterate", "update", "update_bound", "pruning", "depruning", "update_w", "update_Z", "update_b",
                 "update_tau", "update_alpha", "update_x", "update_xi"):
        assert callable(getattr(m, meth))
    assert fav.HyperParameters(2).tau_a == [1e-14, 1e-14]


def test_host_mirror_draws_the_same_initial_parameters_as_the_oracle():
    from oracle import vae_torch as vt
    mod = _load("condition_a_VAE", "celeba_deepvanilla_vae")
    torch.manual_seed(0)
    vae = mod.ImgVAE(dimx=20, channels=3, lr=1e-3, h=64, w=64, dataset="x")
    torch.manual_seed(0)
    P, B = vt.init_state(20, 3)
    sd = vae.state_dict()
    assert len(sd) == 78 and sum(p.numel() for p in vae.parameters()) == 12833003
    for k, v in P.items():
        assert torch.equal(sd[k], v.detach()), k
    for k, v in B.items():
        assert torch.equal(sd[k], v), k
    assert isinstance(vae.optim, torch.optim.Adam) and vae.optim.defaults["lr"] == 1e-3


def test_compute_without_a_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from favae_hip import lib
    mod = _load("condition_a_VAE", "celeba_deepvanilla_vae")
    vae = mod.ImgVAE(dimx=4, channels=3, lr=1e-3, dataset="x")
    with pytest.raises(lib.FavaeHipError):
        vae.update_x(np.zeros((2, 3, 64, 64), dtype=np.float32))
    fav = _load("condition_a_VAE", "favae")
    m = fav.SSHIBA(2, 1)
    with pytest.raises(lib.FavaeHipError):
        m.fit(m.struct_data(np.zeros((4, 3)), "reg"), m.struct_data(np.zeros((4, 2)), "mult"), max_iter=1, pretrained=0)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), os.path.join(root, f)


def test_lazy_dict_materialises_and_invalidates():
    from favae_hip.sshiba import LazyDict
    d = LazyDict(mean=np.arange(6.0).reshape(3, 2), LH=0)
    t = d.dev("mean", "cpu")
    assert torch.equal(t, torch.arange(6.0, dtype=torch.float64).reshape(3, 2))
    d.set_dev("mean", t * 2)
    assert np.array_equal(d["mean"], np.arange(6.0).reshape(3, 2) * 2)
    d["mean"] = np.ones((1, 1))
    assert d.dev("mean", "cpu").shape == (1, 1)
    import pickle
    back = pickle.loads(pickle.dumps(d))
    assert isinstance(back, LazyDict) and np.array_equal(back["mean"], np.ones((1, 1))) and back["LH"] == 0


def test_shard_bounds_partition_every_row_once():
    from favae_hip import dist
    for n in (0, 1, 7, 64, 202599):
        for ws in (1, 2, 3, 8):
            b = [dist.shard_bounds(n, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _gloo_worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=ws)
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    from favae_hip import dist
    from oracle import fa_numpy as fo
    rs = np.random.RandomState(5)
    N, d, K = 1001, 7, 4
    X, Z = rs.randn(N, d), rs.randn(N, K)
    lo, hi = dist.shard_bounds(N)
    Xs, Zs = X[lo:hi], Z[lo:hi]
    # the statistics every rank reduces: X^T [Z | 1], Z^T [Z | 1]  (CPU stand-in for fa_atb + RCCL all-reduce)
    xtz1 = torch.from_numpy(Xs.T @ np.hstack([Zs, np.ones((hi - lo, 1))]))
    ztz1 = torch.from_numpy(Zs.T @ np.hstack([Zs, np.ones((hi - lo, 1))]))
    dist.sum_(xtz1)
    dist.sum_(ztz1)
    full_x = X.T @ np.hstack([Z, np.ones((N, 1))])
    ok = np.allclose(xtz1.numpy(), full_x, rtol=1e-12, atol=1e-12) and np.allclose(ztz1.numpy(), Z.T @ np.hstack([Z, np.ones((N, 1))]),
                                                                                rtol=1e-12, atol=1e-12)
    # a W update computed from the reduced statistics equals the oracle's un-sharded update
    b, tau = rs.randn(1, d) * 0.1, 1.3
    G = rs.randn(K, K)
    wcov = np.linalg.inv(np.eye(K) + G.T @ G)
    W_sh = tau * (xtz1.numpy()[:, :K] - b.T @ ztz1.numpy()[:, K][None, :]) @ wcov
    W_full = tau * np.linalg.multi_dot([(X - b).T, Z, wcov])
    ok = ok and np.allclose(W_sh, W_full, rtol=1e-10, atol=1e-12)
    # gradient averaging and replicated-state broadcast
    g = torch.full((5,), float(rank + 1))
    dist.mean_(g)
    ok = ok and torch.allclose(g, torch.full((5,), (ws + 1) / 2))
    w = torch.full((3,), float(rank))
    dist.broadcast_(w)
    ok = ok and torch.equal(w, torch.zeros(3))
    ok = ok and dist.sum_scalar(hi - lo) == N and dist.all_equal(3) and not dist.all_equal(rank)
    # the statistics of one phase packed into one all-reduce give the per-statistic sums
    a, b2, c = torch.full((3, 4), rank + 1.0, dtype=torch.float64), torch.arange(5, dtype=torch.float64) * (rank + 1), torch.ones(1, 2, dtype=torch.float64)
    ref = [t.clone() for t in (a, b2, c)]
    for t in ref:
        dist.sum_(t)
    out = dist.sum_many_([a, b2, c])
    ok = ok and all(o is t and torch.equal(t, r) for o, t, r in zip(out, (a, b2, c), ref))
    ok = ok and dist.min_scalar(rank) == 0 and dist.max_scalar(rank) == ws - 1
    # column statistics of the observed entries over all shards == np.nanmean / np.nanstd of the whole array (the imputation
    # statistics of semi-supervised views, cA/favae.py:251); a column without an observed entry anywhere comes back as NaN
    A = rs.randn(N, 5) * 3 + 1
    A[rs.rand(N, 5) < 0.3] = np.nan
    A[:, 4] = np.nan
    A[lo:hi, 3] = np.nan if rank == 0 else A[lo:hi, 3]      # observed on one rank only
    cnt, mu, sd = dist.masked_col_stats(A[lo:hi], ~np.isnan(A[lo:hi]))
    full = A.copy()
    full[:dist.shard_bounds(N, ws, 0)[1], 3] = np.nan
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")     # (the all-NaN column)
        fm, fs = np.nanmean(full, axis=0), np.nanstd(full, axis=0)
    ok = ok and np.array_equal(cnt, (~np.isnan(full)).sum(0)) and np.allclose(mu[:4], fm[:4], rtol=1e-12) and np.allclose(sd[:4], fs[:4], rtol=1e-12)
    ok = ok and np.isnan(mu[4]) and np.isnan(sd[4]) and cnt[4] == 0
    q.put((rank, bool(ok)))
    td.destroy_process_group()


def test_two_rank_sharded_statistics_match_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_checkpoint_file_round_trips_with_the_oracle(tmp_path):
    """The restore_weights file of the reference (cA/celeba_deepvanilla_vae.py:300-314: {epoch, elbo, state_dict, optimizer} at
    ./checkpoints/<dataset>_celeba_vanilla_<latentspace_lr>.pt): a file written by ImgVAE.save_checkpoint loads into the
    oracle's module + torch optimiser, a file written from the oracle's state loads into ImgVAE; tensors only
    (torch.load(weights_only=True)).  No GPU: the nn.Module / torch.optim side of ImgVAE is what the file is made from."""
    import torch
    sys.path.insert(0, ROOT)
    from favae_hip.imgvae import ImgVAE
    from oracle import vae_torch as vt
    torch.manual_seed(3)
    vae = ImgVAE(dimx=8, channels=3, lr=1e-3, h=64, w=64, dataset="ck", latentspace_lr=0.5)
    assert vae.checkpoint_path() == "./checkpoints/ck_celeba_vanilla_0.5.pt"
    g = torch.Generator().manual_seed(1)
    for i, p in enumerate(vae.parameters()):                      # an optimiser state as after 7 steps
        vae.optim.state[p] = {"step": torch.tensor(7.0), "exp_avg": torch.randn(p.shape, generator=g) * 1e-3,
                              "exp_avg_sq": torch.rand(p.shape, generator=g) * 1e-6}
    vae.elbo_training = [-123.5]
    path = vae.save_checkpoint(str(tmp_path / "a.pt"), epoch=4)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "elbo", "state_dict", "optimizer"} and ck["epoch"] == 4 and ck["elbo"] == -123.5
    ref = vt.OracleImgVAE(dimx=8, channels=3, lr=1e-3, dataset="ck")
    assert set(ref.state_dict()) <= set(ck["state_dict"])          # same parameter / buffer names as the reference module
    ref.load_state_dict(ck["state_dict"])
    ref.optim.load_state_dict(ck["optimizer"])
    mine = vae.state_dict()
    for k, v in ref.state_dict().items():
        assert torch.equal(v, mine[k]), k
    for p_ref, p_mine in zip([ref.P[k] for k in ref.names], vae.parameters()):
        assert torch.equal(ref.optim.state[p_ref]["exp_avg"], vae.optim.state[p_mine]["exp_avg"])
        assert float(ref.optim.state[p_ref]["step"]) == 7.0
    # the other way: the oracle's state written in the reference layout
    torch.manual_seed(5)
    ref2 = vt.OracleImgVAE(dimx=8, channels=3, lr=1e-3, dataset="ck")
    for k in ref2.names:
        ref2.optim.state[ref2.P[k]] = {"step": torch.tensor(2.0), "exp_avg": torch.full_like(ref2.P[k], 0.25),
                                       "exp_avg_sq": torch.full_like(ref2.P[k], 0.5)}
    sd = ref2.state_dict()
    torch.save({"epoch": 2, "elbo": -1.0, "state_dict": sd, "optimizer": ref2.optim.state_dict()}, str(tmp_path / "b.pt"))
    ep, elbo = vae.load_checkpoint(str(tmp_path / "b.pt"))
    assert (ep, elbo) == (2, -1.0)
    now = vae.state_dict()
    for k, v in ref2.state_dict().items():
        assert torch.equal(v, now[k]), k
    p0 = next(iter(vae.parameters()))
    assert float(vae.optim.state[p0]["step"]) == 2.0 and float(vae.optim.state[p0]["exp_avg"].flatten()[0]) == 0.25
    sys.path.insert(0, os.path.join(PKG, "domain_adaptation"))
    import celeba2cartoon_vae
    assert celeba2cartoon_vae.ImgVAE(dimx=8, channels=3, dataset="d", latentspace_lr=1).checkpoint_path() == "./checkpoints/d81.pt"
