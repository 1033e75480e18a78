"""Host-side mirror of the reference coordinator `favae.SSHIBA` with the factor-analysis state resident on the GPU.

Same constructor / struct_data / fit / fit_iterate / update / update_bound / pruning / predict surface and the same
public attributes (q_dist.{Z,W,b,alpha,tau}, X, t, L, img_vae, total_elbo, recon_loss, kl_q_p, wandb_metrics) as
condition_a_VAE/favae.py (variant 'A'), domain_adaptation/favae.py ('B') and transfer_learning/favae.py ('C').
The coordinator decides *what* runs in which order (reference :743-902) and keeps O(K)-sized bookkeeping; every
N-sized expression and every K x K factorisation runs in libfavae_hip.so (csrc/fa.hip).  No CPU fallback.

Per-sample arrays (X.mean, X.cov, Z.mean, xi, t) live on the device; the dict entries the result scripts index are
materialised as float64 NumPy arrays on access.
"""
import copy
import math
import sys

import numpy as np
import torch

from . import dist
from . import fa
from . import lib as L

EPS = sys.float_info.epsilon
F64 = torch.float64


class LazyDict(dict):
    """dict whose large entries live on the device; host NumPy copies are made on first access after a change."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._dev = {}

    def set_dev(self, key, tensor):
        self._dev[key] = tensor
        dict.__setitem__(self, key, None)

    def dev(self, key, device=None):
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(np.ascontiguousarray(dict.__getitem__(self, key), dtype=np.float64)).to(device)
        return self._dev[key]

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if v is None and key in self._dev:
            d = self._dev[key]
            v = [t.cpu().numpy() if t is not None else np.nan for t in d] if isinstance(d, list) else d.cpu().numpy()
            dict.__setitem__(self, key, v)
        elif v is None and key in getattr(self, "_lazy", {}):       # assembled from device statistics on first access
            v = self._lazy[key]()
            dict.__setitem__(self, key, v)
        return v

    def __setitem__(self, key, value):
        self._dev.pop(key, None)
        getattr(self, "_lazy", {}).pop(key, None)
        self._touched = True          # a caller assigned an entry by hand (cA/condition_a_pretrainedvae.py:94-95)
        dict.__setitem__(self, key, value)

    def __getstate__(self):
        out = {}
        for k in self.keys():
            if isinstance(k, str) and k.startswith("_"):      # device-side by-products (log-determinants of the last inverse)
                continue
            v = self[k]
            out[k] = [v[i] for i in range(len(v))] if isinstance(v, _PredList) else v
        return out

    def __setstate__(self, st):
        self._dev = {}
        self.update(st)

    def __reduce__(self):
        return (LazyDict, (), self.__getstate__())


class HyperParameters(object):
    """reference :1856-1881"""

    def __init__(self, m):
        self.alpha_a, self.alpha_b = [1e-14] * m, [1e-14] * m
        self.tau_a, self.tau_b = [1e-14] * m, [1e-14] * m
        self.gamma_a, self.gamma_b = [2] * m, [1] * m
        self.xi = []


class Qdistribution(object):
    """reference :1883-2063 (non-sparse): containers + the two mean helpers; XS = X* of semi-supervised 'reg' views."""

    def __init__(self, X, n, n_max, d, Kc, m, hyper):
        self.n, self.n_max, self.d, self.Kc, self.m, self.X = n, n_max, d, Kc, m, X
        self.alpha = [LazyDict(a=hyper.alpha_a[i], b=hyper.alpha_b[i] * np.ones(Kc), LH=None, ElogpWalp=None) for i in range(m)]
        self.tau = [dict(a=hyper.tau_a[i], b=hyper.tau_b[i] * np.ones(1), LH=None, ElogpXtau=None) for i in range(m)]
        self.gamma = [dict(a=hyper.gamma_a[i], b=hyper.gamma_b[i] * np.ones(d[i]), LH=None) for i in range(m)]
        blank = lambda: dict(mean=None, cov=None, prodT=None, LH=0, Elogp=0)
        self.W = [LazyDict(blank()) for _ in range(m)]      # K-sized state is device-resident between the updates too;
        self.b = [LazyDict(blank()) for _ in range(m)]      # the host arrays the result scripts index are made on access
        self.XS, self.tS = [LazyDict(blank()) for _ in range(m)], [blank() for _ in range(m)]
        self.tc = {}
        self.Z = LazyDict(blank())
        self.xi = [None] * m

    def alpha_mean(self, m):
        return self.alpha[m]["a"] / self.alpha[m]["b"]

    def tau_mean(self, m):
        return self.tau[m]["a"] / self.tau[m]["b"]

    def gamma_mean(self, m):
        return self.gamma[m]["a"] / self.gamma[m]["b"]


class SSHIBA(object):
    VARIANT = "A"          # set by the thin favae.py modules
    img_vae_module = None  # the plugin module (`import celeba_deepvanilla_vae as img_vae`, reference :20)

    def __init__(self, Kc=2, prune=0, fs=0, SS_sep=0, hyper=None, X_init=None, b_init=None, Z_init=None, W_init=[None],
                 alpha_init=None, tau_init=None, gamma_init=None, area_mask=None, latentspace_lr=1):
        self.Kc, self.prune, self.fs, self.SS_sep = int(Kc), int(prune), int(fs), SS_sep
        self.hyper, self.X_init, self.b_init, self.Z_init, self.W_init = hyper, X_init, b_init, Z_init, W_init
        self.alpha_init, self.tau_init, self.gamma_init = alpha_init, tau_init, gamma_init
        self.latentspace_lr = latentspace_lr
        for name, v in (("fs", fs), ("X_init", X_init), ("b_init", b_init), ("Z_init", Z_init), ("alpha_init", alpha_init),
                        ("tau_init", tau_init), ("gamma_init", gamma_init)):
            if v:
                raise NotImplementedError("SSHIBA(%s=...) is outside the FA-VAE training hot path (SURVEY.md 2 row 13)" % name)

    # ------------------------------------------------------------------------------------------------------------
    def struct_data(self, X, method, sparse=0, V=None, kernel=None, sig=0, sparse_fs=0, center=1, mask=None, deg=None,
                    latent_dim=None, lr=None, epochs=None, dataset=None):
        """reference :585-621"""
        if V is not None:
            kernel = "rbf" if kernel is None else kernel.lower()
        return dict(data=X, sparse=sparse, sparse_fs=sparse_fs, method=method, mask=mask, SV=V, kernel=kernel, sig=sig,
                    center=center, deg=deg, latent_dim=latent_dim, lr=lr, epochs=epochs, dataset=dataset)

    # ---- device helpers ------------------------------------------------------------------------------------------
    def _device_state(self):
        """Device handle + scratch buffers; built by fit, and again on first use by a model that came out of a pickle (the
        result scripts un-pickle a fitted model and call predict, DA/results_celeba2cartoon.py:25-34)."""
        import os
        if not torch.cuda.is_available():
            raise L.FavaeHipError("SSHIBA needs a GPU: the HIP path has no CPU fallback")
        self.dev = torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(self.dev)
        self.sc = fa.Scratch(self.dev)
        self.scal = torch.zeros(4, dtype=F64, device=self.dev)

    def __getattr__(self, name):          # only reached when the attribute is missing: device state dropped by pickling
        if name in ("dev", "sc", "scal") and "q_dist" in self.__dict__:
            self._device_state()
            return self.__dict__[name]
        if name in ("_pending_scal", "_pending_mult", "_pending_info"):      # nothing is pending on a model that was just un-pickled
            self.__dict__[name] = [] if name == "_pending_info" else {}
            return self.__dict__[name]
        raise AttributeError(name)

    def _resume_device_stats(self):
        """After un-pickling: the reduced statistics the updates read (column sums, traces, Z^T [Z | 1]) and the device copies
        of xi / the semi-supervised masks are rebuilt from the pickled state, so fit_iterate can go on where it stopped."""
        q = self.q_dist
        q.xi = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.dev) if isinstance(x, np.ndarray) else x
                for x in q.xi]
        if any(self.SS):
            self._ss_mask_dev = {m: torch.from_numpy(self.SS_mask[m] if self.method[m] == "reg" else self.SS_mask[m].astype(np.uint8)).to(self.dev)
                                 for m in range(self.m) if self.SS[m]}
        self._xsum, self._xtx1, self._covsum = {}, {}, {}
        self._pending_info, self._pending_scal, self._pending_mult = [], {}, {}
        for m in range(self.m):
            prod = dict.__getitem__(self.X[m], "prodT")
            self._refresh_xstats(m)
            dict.__setitem__(self.X[m], "prodT", prod)
            self.X[m]._touched = False
        self._refresh_zstats()

    def _up(self, a):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.dev)

    def _tau(self, m):
        return float(np.ravel(self.q_dist.tau_mean(m))[0])

    def _refresh_xstats(self, m):
        """X^T [X | 1] and sum_n cov -> X.prodT (:785/818/1213), 1^T X, tr<X^T X>."""
        X = self.X[m]
        Xd = X.dev("mean", self.dev)
        d = self.d[m]
        stats = [fa.atb(Xd, Xd, True, self.sc)]
        with_cov = self.method[m] != "reg" or self.SS[m]
        if with_cov:
            stats.append(fa.atb(X.dev("cov", self.dev), None, True, self.sc))
        dist.sum_many_(stats)                              # one all-reduce for the statistics of this view
        xtx1 = stats[0]
        self._xsum[m] = xtx1[:, d].contiguous()
        # X.prodT = X^T X + diag(sum cov) and its trace stay on the device (the tau update takes them from there); the host
        # array is assembled when somebody reads X['prodT']
        self._xtx1[m] = xtx1
        self._covsum[m] = stats[1].reshape(-1).contiguous() if with_cov else None
        dict.__setitem__(X, "prodT", None)
        X._dev.pop("prodT", None)
        X._lazy = getattr(X, "_lazy", {})
        cs = self._covsum[m]
        X._lazy["prodT"] = lambda xtx1=xtx1, cs=cs, d=d: (xtx1[:, :d].cpu().numpy() + (np.diag(cs.cpu().numpy()) if cs is not None else 0.0))

    def _refresh_zstats(self, also=()):
        """Z^T [Z | 1] summed over the shards; `also`: further local statistics of the same phase, reduced in the same
        all-reduce (in place)."""
        q = self.q_dist
        K = q.Kc
        ztz1 = fa.atb(q.Z.dev("mean", self.dev), q.Z.dev("mean", self.dev), True, self.sc)
        dist.sum_many_([ztz1] + list(also))
        self._zsum = ztz1[:, K].contiguous()
        return ztz1[:, :K].contiguous()

    # ---- fit (:84-379) -------------------------------------------------------------------------------------------
    def fit(self, *args, **kwargs):
        if not torch.cuda.is_available():
            raise L.FavaeHipError("SSHIBA.fit needs a GPU: the HIP path has no CPU fallback")
        self._device_state()
        n = [int(a["data"].shape[0]) for a in args]
        self.n_loc = int(np.max(n))                       # rows held by this rank (the views passed in are this rank's shard)
        self.n_max = int(round(dist.sum_scalar(self.n_loc, self.dev)))      # global N of the formulas
        self.n, self.d, self.sparse, self.sparse_fs, self.center, self.method = [], [], [], [], [], []
        self.X, self.t, self.SS, self.SS_mask, self.area_mask = [], {}, [], {}, []
        self.V, self.k, self.sig, self.deg, self.sparse_K, self.kernel = {}, {}, {}, {}, {}, {}
        self.img_vae = [None] * len(n)
        self.cat_vae = [None] * len(n)
        self.wandb_obj = [None] * len(n)
        self.total_elbo, self.kl_q_p, self.recon_loss, self.kl_q_normal = [], [], [], []
        self.keep_training, self.test_acc = [], []
        self._xsum, self._trxx, self._xtx1, self._covsum = {}, {}, {}, {}
        self._pending_info, self._pending_scal, self._pending_mult = [], {}, {}
        for m, arg in enumerate(args):
            meth = arg["method"]
            if arg.get("SV") is not None or arg.get("sparse") or meth not in ("reg", "img", "mult"):
                raise NotImplementedError("view %d: kernel / sparse / '%s' views are outside the FA-VAE hot path" % (m, meth))
            self.n.append(int(arg["data"].shape[0]))
            if self.n[m] != self.n_loc and (meth != "reg" or dist.active()):
                raise NotImplementedError("views with fewer rows: only 'reg' views, single process (pass NaN rows instead)")
            if meth == "reg":
                self.d.append(int(arg["data"].shape[1]))
            elif meth == "img":
                self.d.append(int(arg["latent_dim"]))
            else:
                if len(arg["data"].shape) < 2:
                    arg["data"] = arg["data"][:, np.newaxis]
                self.d.append(int(arg["data"].shape[1]))
            self.sparse.append(arg["sparse"]); self.sparse_fs.append(arg["sparse_fs"]); self.center.append(arg["center"])
            self.method.append(meth); self.area_mask.append(arg["mask"]); self.SS.append(False)
            d = self.d[m]
            mn = np.random.normal(0.0, 1.0, self.n_loc * d).reshape(self.n_loc, d)                    # :179
            info = LazyDict(data=np.random.normal(0.0, 1.0, self.n_loc * d).reshape(self.n_loc, d), mean=mn, cov=mn ** 2,
                            prodT=np.dot(mn.T, mn), LH=0, Elogp=0, sumlogdet=0, epochs=20)
            self.X.append(info)
            data = arg["data"] if meth == "img" else copy.deepcopy(arg["data"])      # images are read-only here
            if meth == "reg":                                                                        # :236-257
                info["data"] = data
                mean = np.copy(data).astype(np.float64)
                if self.n_loc > self.n[m]:
                    mean = np.vstack((mean, np.nan * np.ones((self.n_loc - self.n[m], d))))
                mask = np.isnan(mean)
                self.SS_mask[m] = mask
                if dist.sum_scalar(float(mask.sum()), self.dev) > 0:                                 # semi-supervised view
                    self.SS[m] = True
                    # the imputation draws use the column statistics of the whole data set (:251): over all shards when sharded
                    gobs, gmean, gstd = dist.masked_col_stats(mean, ~mask, self.dev) if dist.active() else (None, None, None)
                    for dd in range(d):
                        cnt = int(np.sum(mask[:, dd]))
                        if (gobs[dd] == 0) if dist.active() else (cnt == mask.shape[0]):
                            mean[mask[:, dd], dd] = np.random.normal(0.0, 1.0, cnt)
                        elif dist.active():
                            mean[mask[:, dd], dd] = np.random.normal(gmean[dd], gstd[dd], cnt)
                        else:
                            mean[mask[:, dd], dd] = np.random.normal(np.nanmean(mean[:, dd], axis=0), np.nanstd(mean[:, dd], axis=0), cnt)
                    cov = np.copy(info["cov"])
                    cov[~mask] = 0.0                                                                 # unobserved entries keep mn**2 (:179)
                    info["mean"], info["cov"] = mean, cov
                    self._ss_mask_dev = getattr(self, "_ss_mask_dev", {})
                    self._ss_mask_dev[m] = torch.from_numpy(mask).to(self.dev)
                    self._ss_count = getattr(self, "_ss_count", {})
                    self._ss_count[m] = float(mask.sum())                                            # (summed over the shards where used)
                else:
                    info["mean"] = mean
                    info["cov"] = np.zeros(mean.shape)
            elif meth == "img":
                self.keep_training.append(True)
                self.t[m] = dict(data=data, epochs=arg["epochs"], LH=0, Elogp=0, sumlogdet=0)
                self.img_vae[m] = self.img_vae_module.ImgVAE(dimx=arg["latent_dim"], channels=data.shape[1], lr=arg["lr"],
                                                             h=data.shape[2], w=data.shape[3], dataset=arg["dataset"],
                                                             latentspace_lr=self.latentspace_lr)
                self.img_vae[m].defer_pull = True      # fit_iterate syncs the nn.Module once, when it returns
            else:                                                                                    # :269-289 (SS_sep = 0)
                tdata = data.astype(float)
                mask = np.isnan(tdata)
                self.SS_mask[m] = mask
                self.t[m] = LazyDict(data=tdata, LH=0, Elogp=0, sumlogdet=0, epochs=20)
                if dist.sum_scalar(float(mask.sum()), self.dev) > 0:                                 # semi-supervised labels
                    if self.SS_sep:
                        raise NotImplementedError("semi-supervised multilabel views: SS_sep=0 only (the reference's SS_sep branch, "
                                                  "cA/favae.py:294-331, appends to the dict self.t and cannot run)")
                    self.SS[m] = True
                    tmean, tcov = np.copy(info["mean"]), np.copy(info["cov"])                        # t = deepcopy(info) (:270)
                    mean, cov = np.copy(info["mean"]), np.copy(info["cov"])
                    tt = np.abs(tdata[~mask] - 0.05)
                    mean[~mask] = np.log(tt / (1 - tt))                                              # :276
                    cov[~mask] = 0
                    if dist.active():     # column statistics of the whole data set (:278-286): over all shards
                        _, omean, ostd = dist.masked_col_stats(mean, ~mask, self.dev)
                        _, tmu, tsd = dist.masked_col_stats(tmean, np.ones_like(mask), self.dev)
                    for dd in range(d):                                                              # :278-280
                        obs, cnt = mean[~mask[:, dd], dd], int(np.sum(mask[:, dd]))
                        mu, sd = (omean[dd], ostd[dd]) if dist.active() else (np.nanmean(obs, axis=0), np.nanstd(obs, axis=0))
                        mean[mask[:, dd], dd] = np.random.normal(mu, sd, cnt)
                        cov[mask[:, dd], dd] = np.abs(np.random.normal(mu, sd, cnt))
                    for dd in range(d):                                                              # :285-286
                        mu, sd = (tmu[dd], tsd[dd]) if dist.active() else (np.nanmean(tmean[:, dd], axis=0), np.nanstd(tmean[:, dd], axis=0))
                        tmean[mask[:, dd], dd] = np.random.normal(mu, sd, int(np.sum(mask[:, dd])))
                    tmean[~mask] = tdata[~mask]
                    tcov[~mask] = 0
                    info["mean"], info["cov"] = mean, cov
                    self.t[m]["mean"], self.t[m]["cov"] = tmean, tcov
                    self._ss_mask_dev = getattr(self, "_ss_mask_dev", {})
                    self._ss_mask_dev[m] = torch.from_numpy(mask.astype(np.uint8)).to(self.dev)
                else:
                    tt = np.abs(tdata - 0.05)
                    info["mean"] = np.log(tt / (1 - tt))                                             # :276
                    info["cov"] = np.zeros(info["mean"].shape)
                    self.t[m]["mean"] = np.copy(tdata)
                    self.t[m]["cov"] = np.zeros(tdata.shape)
        self.keep_training = {m: True for m in range(len(n)) if self.method[m] == "img"}
        self.L, self.mse, self.R2, self.HL, self.AUC, self.AUC_tr, self.ACC, self.ACC_tr = [], [], [], [], [], [], [], []
        self.wandb_metrics = {}
        self.m = len(args)
        if self.hyper is None:
            self.hyper = HyperParameters(self.m)
        q = self.q_dist = Qdistribution(self.X, self.n, self.n_max, self.d, self.Kc, self.m, self.hyper)
        # ---- Qdistribution.init_rnd (:1919-1995), same draw order --------------------------------------------
        for m in range(self.m):
            xi = torch.empty(self.n_loc, self.d[m], dtype=F64, device=self.dev)
            fa.xi_init(self.X[m].dev("mean", self.dev), self.X[m].dev("cov", self.dev), xi)
            q.xi[m] = xi
        q.Z["mean"] = np.random.normal(0.0, 1.0, self.n_loc * self.Kc).reshape(self.n_loc, self.Kc)
        q.Z["cov"] = np.eye(self.Kc)
        q.Z["prodT"] = q.Z["cov"] + self.n_max * q.Z["cov"]
        self._refresh_zstats()
        for m in range(self.m):
            if self.SS[m]:                                                                           # :1963-1970, same draws
                q.tc[m] = np.random.rand(self.n_loc, self.d[m])
                q.tS[m]["mean"] = np.random.randint(2, size=[self.n_loc, self.d[m]])
                q.tS[m]["cov"] = np.eye(self.d[m])
                q.tS[m]["sumlogdet"] = 0
                q.XS[m]["mean"] = np.random.normal(0.0, 1.0, self.n_loc * self.d[m]).reshape(self.n_loc, self.d[m])
                q.XS[m]["cov"] = np.zeros((self.n_loc, self.d[m]))
            W = q.W[m]
            W["mean"] = np.random.normal(np.zeros((self.d[m], self.Kc)),
                                         1 / (np.repeat(q.alpha_mean(m).reshape(1, self.Kc), self.d[m], axis=0)))
            if dist.active():                              # replicated state must start identical on every rank
                W["mean"] = dist.broadcast_(self._up(W["mean"])).cpu().numpy()
            W["cov"] = np.dot(W["mean"].T, W["mean"])
            W["prodT"] = np.dot(W["mean"].T, W["mean"]) + self.Kc * W["cov"]
            self._refresh_xstats(m)
            self.update_b(m, n=self.n[m] if self.n[m] != self.n_loc else None)                       # :1986-1988 (n[m], not n_max)
        for X in self.X:
            X._touched = False
        if self.VARIANT == "A":
            if not kwargs["pretrained"]:                                                             # :378 (KeyError if absent)
                self.fit_iterate(**kwargs)
        else:
            self.fit_iterate(**kwargs)

    # ---- fit_iterate (:383-451) ----------------------------------------------------------------------------------
    def fit_iterate(self, *args, **kwargs):
        try:
            return self._fit_iterate(*args, **kwargs)
        finally:                       # the image models' nn.Module parameters / optimiser state follow the executor once
            for vae in self.img_vae:
                if vae is not None:
                    vae.sync()
            if "q_dist" in self.__dict__ and "_pending_scal" in self.__dict__:
                self._sync_scalars()
                self._flush_info()

    def _fit_iterate(self, max_iter=int(1e3), pruning_crit=1e-6, tol=1e-5, feat_crit=1e-6, perc=False, verbose=0, Y_tst=[None],
                     X_tst=[None], X_tr=[None], HL=0, AUC=0, ACC=0, mse=0, R2=0, m_in=[0], m_out=1, steps=50, nnd=None,
                     pretrained=0, mc_it=1, Y_ss=None, store=False, xga=None, tgx=None, lr=None):
        q = self.q_dist
        for vae in self.img_vae:           # also a model injected after fit(pretrained=1): one sync when fit_iterate returns
            if vae is not None:
                vae.defer_pull = True
        for i in range(max_iter):
            self.update(iter_count=i)
            if self.prune:
                self.depruning(1e-15)
                self.L.append(self.update_bound())
                self.pruning(pruning_crit)
                if q.Kc == 0:
                    print("\nThere are no representative latent factors, no structure found in the data.")
                    return
            else:
                self.L.append(self.update_bound())
            if verbose:
                text = "Nothing" if len(self.L) < 2 else (" " if self.L[-1] > self.L[-2] else "LOWER!!!!")
                print("\rIteration %d Lower Bound %.1f K %4d %s" % (len(self.L), self.L[-1], q.Kc, text))
            if (len(self.L) > 100) and (abs(1 - np.mean(self.L[-101:-1]) / self.L[-1]) < tol):
                if verbose:
                    print("\nModel correctly trained. Convergence achieved")
                return
            self.wandb_metrics["SSHIBA ELBO"] = float(self.L[-1])
            self.wandb_metrics["K dimensions"] = q.Kc

    # ---- update (:743-902) ---------------------------------------------------------------------------------------
    def update(self, Y_tst=[None], X_tst=[None], X_tr=[None], Y_ss=None, HL=0, AUC=0, ACC=0, mse=0, R2=0, m_in=[0], m_out=1,
               steps=100, mc_it=1, store=True, iter_count=0):
        if "_xsum" not in self.__dict__:
            self._resume_device_stats()
        for m in range(self.m):
            if getattr(self.X[m], "_touched", False):      # X[m]['mean'] / ['cov'] were assigned from outside: the column sums
                self._refresh_xstats(m)                    # and traces the tau / b updates read follow them
                self.X[m]._touched = False
        # X_m^T [Z | 1] of every view in one all-reduce (Z does not change between the W updates)
        Zd = self.q_dist.Z.dev("mean", self.dev)
        xtz = dist.sum_many_([fa.atb(self.X[m].dev("mean", self.dev), Zd, True, self.sc) for m in range(self.m)])
        self._update_w_views(list(range(self.m)), xtz)      # the M Cholesky inverses: one launch
        self.update_Z()
        for m in range(self.m):
            meth = self.method[m]
            if meth == "reg":
                if self.SS[m]:
                    self.update_xs(m)                                                                # :775-783
                self.update_tau(m)
                self.update_b(m)
            if meth == "img":
                self._update_img(m, iter_count)
            elif meth == "mult":
                for _ in range(2):
                    self.update_x(m)          # update_x + update_xi fused (:856-857)
                    if self.SS[m]:
                        self.update_t(m)      # :858-861
                self._refresh_xstats(m)
                self.update_tau(m)
            self.update_b(m)
            self.update_alpha(m)

    def _schedule(self, m, iter_count):
        if self.VARIANT == "A":                                                                      # cA :792-800
            if iter_count == 0:
                return 50, 1
            if iter_count < 3:
                return 5, 1
            self.keep_training[m] = False
            return 5, 5
        if iter_count == 0:                                                                          # DA :808-810
            return 50, 0.1
        if iter_count == 1:
            return 20, 0.5
        self.keep_training[m] = False
        return 20, 1

    def _update_img(self, m, iter_count):
        q, vae = self.q_dist, self.img_vae[m]
        self._sync_scalars()            # the VAE plugin reads b['cov'] / b['prodT'] on the host when it samples the prior
        epochs, beta = self._schedule(m, iter_count)
        if getattr(self, "epochs_override", None) is not None:
            epochs = self.epochs_override[min(iter_count, len(self.epochs_override) - 1)]
        if self.VARIANT == "A":
            # :802 (the moment prior of celeba_deepvanilla_vae never looks at the covariance: not fetched for it)
            Z_in = {"mean": q.Z.dev("mean_pred%d" % m, self.dev),
                    "cov": q.Z["covs_pred"][m] if getattr(vae, "PRIOR", "moment") != "moment" else None}
        else:
            Z_in = {"mean": q.Z.dev("mean", self.dev), "cov": q.Z["cov"]}
        img = self.t[m]["data"]
        trained = False
        if self.keep_training[m]:
            kw = dict(img=img, Z=Z_in, W=q.W[m], b=q.b[m], tau=q.tau_mean(m), epochs=epochs, beta=beta)
            if self.VARIANT == "A":
                kw["favae"] = True                                                                   # cA :806
            # variant C (TL :812) does not pass `favae`, so betaVAE.trainloop falls back to the N(0,I) prior
            self.keep_training[m] = vae.trainloop(**kw)
            trained = True
        elif iter_count % 10 == 0:
            kw = dict(img=img, Z=Z_in, W=q.W[m], b=q.b[m], tau=q.tau_mean(m), epochs=epochs)          # :821 / DA :827
            vae.trainloop(**kw)
            trained = True
        if trained:
            ds = str(vae.dataset)
            self.wandb_metrics[ds + " ELBO"] = float(vae.elbo_training[-1])
            self.wandb_metrics[ds + " Gaussian LogLikelihood"] = float(-vae.reconstruc_during_training[-1])
            self.wandb_metrics[ds + " KL(Q||P)"] = float(vae.KL_QandP[-1])
            self.total_elbo.append(vae.elbo_training)
            self.recon_loss.append(vae.reconstruc_during_training)
            self.kl_q_p.append(vae.KL_QandP)
            vae.update_x(img=img)                                                                    # :817
            self.X[m].set_dev("mean", vae.x_mean_dev)
            self.X[m].set_dev("cov", vae.x_cov_dev)
            self._refresh_xstats(m)
        self.update_tau(m)
        self.update_b(m)
        # diagnostics (:839-851): reconstruction R^2 (one np.random.randn draw) and the zero-input prediction R^2
        r2 = vae.reconstruction_r2(self.X[m].dev("mean", self.dev), self.X[m].dev("cov", self.dev))
        self.wandb_metrics[str(vae.dataset) + " R2 reconstruction"] = r2
        if self.VARIANT == "A" and self.m > 1 and self.method[1] == "mult":
            self.wandb_metrics[str(vae.dataset) + " r2_acc in the second view"] = self._zero_input_r2()

    def _zero_input_r2(self):
        """predict([0],[1], struct_data(X['data'],'vae')) of cA :846-848: a 'vae' view contributes nothing to Z_mean
        (:1402-1414), so the prediction is sigmoid(b (1+pi/8 var)^-1/2) per label (:1469-1477); R^2 from column sums."""
        q = self.q_dist
        W0, W1 = q.W[0].dev("mean", self.dev), q.W[1].dev("mean", self.dev)
        ww = fa.small_gemm(W0, W0, transA=True)
        (Zc,), _, _ = fa.chol_inv_batched([dict(terms=[(self._tau(0), ww)], diag_add=1.0)], q.Kc, self.sc)     # :1370-1373
        var = fa.small_gemm(fa.small_gemm(W1, Zc), W1, transB=True)
        d = self.d[1]
        if getattr(self, "_t1_stats", None) is None or self.SS[1]:     # T^T [T | 1]: the labels only change when some are unobserved
            T = self.t[1].dev("mean", self.dev)
            self._t1_stats = dist.sum_(fa.atb(T, T, True, self.sc))
        flat = torch.cat([var.diagonal(), q.b[1].dev("mean", self.dev).reshape(-1), self._t1_stats.reshape(-1)]).cpu().numpy()   # one read-back
        var_d = 1.0 / self._tau(1) + flat[:d]
        b1 = flat[d:2 * d]
        st = flat[2 * d:].reshape(d, d + 1)
        pred = np.exp(-np.log(1 + np.exp(-(b1 * (1 + math.pi / 8 * var_d) ** (-0.5)))))
        n = self.n_max
        sy, syy = st[:, d], np.diag(st[:, :d])
        num = syy - 2 * pred * sy + n * pred ** 2
        den = syy - sy ** 2 / n
        out = np.ones(d)
        nz = den != 0
        out[nz] = 1 - num[nz] / den[nz]
        out[(~nz) & (num != 0)] = 0.0
        return float(out.mean())

    # ---- predict (cA :1336-1483 / :1485-1570, DA :1345-1478) ------------------------------------------------------
    def _predict_latent(self, m_in, args):
        """Posterior of Z given the input views (:1363-1423).  As in the reference only dict inputs of method 'img' and
        'mult' contribute (a list input fails at :1378, other dict views add nothing).  Z_mean stays on the device."""
        q = self.q_dist
        K = q.Kc
        if not isinstance(args[0], dict):
            raise TypeError("list indices must be integers or slices, not str")      # what :1378 raises for list inputs
        for arg in args:
            if arg.get("SV") is not None:
                raise NotImplementedError("kernel views are outside the FA-VAE hot path")
        n_pred = int(np.shape(args[0]["data"])[0])
        self._sync_scalars()
        aux = self._up(np.eye(K))
        for m in m_in:
            Wm = q.W[m].dev("mean", self.dev)
            aux = fa.axpby(1.0, aux, 1.0, fa.small_gemm(Wm, Wm, transA=True, alpha=self._tau(m)))
        Zc, _, _, ok = fa.chol_inv(aux, self.sc)
        X_sample = None
        if not ok:
            print("Cov Z is not invertible")
            return n_pred, None, np.nan, X_sample
        terms, cvec = [], None
        for m, arg in enumerate(args):
            mi = m_in[m]
            Xs = None
            if arg["method"] == "img":
                X_mean, X_cov = self.img_vae[mi].update_x(arg["data"])
                X_sample = X_mean + np.random.randn() * np.sqrt(X_cov)
                Xs = X_sample
            if arg["method"] == "mult":
                Xs = np.log(np.abs((arg["data"] - 0.05)) / (1 - np.abs((arg["data"] - 0.05))))
            if Xs is None:
                continue
            # sum_m tau_m (X_m - b_m) W_m Sigma  =  sum_m X_m P_m + c ,  P_m = tau_m W_m Sigma  (row GEMMs over N)
            P = fa.small_gemm(q.W[mi].dev("mean", self.dev), Zc, alpha=self._tau(mi))
            cvec = fa.small_gemm(q.b[mi].dev("mean", self.dev).reshape(1, -1), P, alpha=-1.0,
                                 beta=0.0 if cvec is None else 1.0, out=cvec)
            terms.append((self._up(Xs), P))
        if not terms:
            Zd = torch.zeros(n_pred, K, dtype=F64, device=self.dev)
        else:
            Zd = torch.empty(n_pred, K, dtype=F64, device=self.dev)
            for i, (Xd, P) in enumerate(terms):
                if i == 0:
                    fa.rows_gemm(Xd, P, cvec.reshape(-1), 1.0, None, 0.0, Zd)
                else:
                    fa.rows_gemm(Xd, P, None, 1.0, None, 0.0, Zd, accumulate=True)
        return n_pred, Zd, Zc.cpu().numpy(), X_sample

    def _predict_outputs(self, m_outs, n_pred, Zd, Z_cov, X_sample, sameimage):
        q = self.q_dist
        if Zd is None:                       # the reference goes on with the Z_mean of an earlier call (or fails)
            Zd = self._up(self.Z_mean)
        self.Z_mean = Zd.cpu().numpy()
        predictions = {}
        if isinstance(m_outs, int):
            m_outs = [m_outs]
        for m_out in m_outs:
            W, b = np.asarray(q.W[m_out]["mean"]), np.asarray(q.b[m_out]["mean"]).reshape(1, -1)
            d = self.d[m_out]
            Wt = self._up(W.T)

            def rows_affine(shift):
                out = torch.empty(n_pred, d, dtype=F64, device=self.dev)
                return fa.rows_gemm(Zd, Wt, self._up(shift.reshape(-1)), 1.0, None, 0.0, out)

            if self.method[m_out] == "reg":
                var_x = q.tau_mean(m_out) ** (-1) * np.eye(d) + np.linalg.multi_dot([W, Z_cov, W.T])
                predictions["output_view" + str(m_out)] = {"mean_x": rows_affine(b).cpu().numpy(), "var_x": var_x}
            if self.method[m_out] == "img":
                vae = self.img_vae[m_out]
                if self.VARIANT == "A" and not sameimage:                                            # cA :1438-1446
                    shift = np.random.normal() * np.sqrt(np.diag(Z_cov))
                    sampleX = rows_affine(shift @ W.T - b)                                           # minus b, as written
                    pred_img = vae.decoder(sampleX.float())
                    predictions = pred_img, self.Z_mean, 0
                else:
                    Z_in = {"mean": self.Z_mean, "cov": Z_cov}
                    pred_img, xcart = vae.predict(Z=Z_in, W=q.W[m_out], b=q.b[m_out], tau=q.tau_mean(m_out))
                    if self.VARIANT == "A":
                        predictions = pred_img, self.Z_mean, xcart
                    else:
                        if X_sample is None:
                            raise NameError("name 'X_sample' is not defined")                      # DA :1450 without an image input
                        predictions = pred_img, X_sample, Z_in, xcart
            elif self.method[m_out] == "mult":
                var_d = 1.0 / self._tau(m_out) + np.einsum("ij,jk,ik->i", W, Z_cov, W)
                scale = self._up((1 + math.pi / 8 * var_d) ** (-0.5))
                mean_t = torch.empty(n_pred, d, dtype=F64, device=self.dev)
                var_t = torch.empty_like(mean_t)
                fa.mult_predict(rows_affine(b), scale, mean_t, var_t)
                predictions["output_view" + str(m_out)] = {"mean_x": mean_t.cpu().numpy(), "var_x": var_t.cpu().numpy()}
        return predictions

    def predict(self, m_in, m_outs, *args):
        n_pred, Zd, Z_cov, X_sample = self._predict_latent(m_in, args)
        return self._predict_outputs(m_outs, n_pred, Zd, Z_cov, X_sample, False)

    def predict_sameimage(self, m_in, m_outs, *args):
        n_pred, Zd, Z_cov, X_sample = self._predict_latent(m_in, args)
        return self._predict_outputs(m_outs, n_pred, Zd, Z_cov, X_sample, True)

    # ---- mean-field updates ----------------------------------------------------------------------------------------
    def update_w(self, m, xtz1=None):
        """:1058-1075 (xtz1: X_m^T [Z | 1] already summed over the shards)"""
        self._update_w_views([m], {m: xtz1})

    def _update_w_views(self, views, xtz):
        """update_w of several views: their K x K inverses (:1062-1065) are one batched launch; the flags stay on the device --
        W is replaced by a kernel that looks at its flag, so "not invertible, not updated" (:1073-1075) needs no read-back here --
        and are reported with those of update_Z."""
        q = self.q_dist
        K = q.Kc
        zp = q.Z.dev("prodT", self.dev)
        taus = {m: self._tau(m) for m in views}
        mats = [dict(terms=[(1.0, zp)], s=q.alpha[m].dev("b", self.dev), s_mul=1.0 / q.alpha[m]["a"], s_sqrt=True,
                     diag_add=1 / taus[m], out_scale=1 / taus[m]) for m in views]
        outs, logdet, info = fa.chol_inv_batched(mats, K, self.sc)
        for i, m in enumerate(views):
            x = xtz[m]
            if x is None:
                x = dist.sum_(fa.atb(self.X[m].dev("mean", self.dev), q.Z.dev("mean", self.dev), True, self.sc))
            W = q.W[m]
            wmean, wprod, wcov = W.dev("mean", self.dev), W.dev("prodT", self.dev), W.dev("cov", self.dev)
            tmp = torch.empty(self.d[m] * K, dtype=F64, device=self.dev)
            fa.w_update(x, self._zsum, q.b[m].dev("mean", self.dev), outs[i], taus[m], wmean, wprod, self.sc, info=info[i:i + 1],
                        wcov_out=wcov, tmp=tmp)
            for k, t in (("mean", wmean), ("prodT", wprod), ("cov", wcov)):
                W.set_dev(k, t)
            W["_logdet_cov"] = logdet[i, 1]
        self._pending_info.append((info, ["Cov W" + str(m) + " is not invertible, not updated" for m in views]))

    def _flush_info(self, extra=None):
        """ONE read-back of the flags of the inverses queued since the last one (the W updates of this iteration and, as `extra`,
        the launch of update_Z); prints the reference's messages; returns the flags of `extra`."""
        pend = self._pending_info
        self._pending_info = []
        ts = [i for i, _ in pend] + ([extra] if extra is not None else [])
        if not ts:
            return []
        flags = torch.cat(ts).cpu().tolist()
        o = 0
        for i, msgs in pend:
            for k, msg in enumerate(msgs):
                if flags[o + k]:
                    print(msg)
            o += i.numel()
        return flags[o:]

    def update_Z(self):
        """cA :961-1000 (variant A) / DA :980-1009 (variants B, C)"""
        q = self.q_dist
        M, K, lr = range(self.m), q.Kc, self.latentspace_lr
        taus = [self._tau(m) for m in M]
        wp = [q.W[m].dev("prodT", self.dev) for m in M]
        # Sigma_Z = (I + sum_m tau_m <W^T W>_m)^-1 and, variant A, the predictive covariances whose `aux` is carried over from
        # view to view (:975-980): M + 1 inverses, one launch, the sums formed while the matrices are loaded
        mats = [dict(terms=[(taus[m], wp[m]) for m in M], diag_add=1.0)]
        if self.VARIANT == "A":
            coef = np.zeros(self.m)
            for m in M:
                coef += np.array([0.0 if j == m else 1.0 for j in M])              # `aux` is carried over (:979)
                terms = [(coef[j] * taus[j], wp[j]) for j in M if coef[j] != 0.0] or [(0.0, wp[0])]
                mats.append(dict(terms=terms, diag_add=1.0))
        outs, logdet, info = fa.chol_inv_batched(mats, K, self.sc)
        flags = self._flush_info(info)                    # the one read-back of the iteration's inverses
        Zcov, covs_pred = outs[0], outs[1:]
        if self.VARIANT == "A":
            q.Z.set_dev("covs_pred", [c if not f else None for c, f in zip(covs_pred, flags[1:])])
        if flags[0]:
            print("Cov Z is not invertible, not updated")
            return
        q.Z.set_dev("cov", Zcov)
        q.Z["_logdet_cov"] = logdet[0, 1]
        Wd = [q.W[m].dev("mean", self.dev) for m in M]
        bd = [q.b[m].dev("mean", self.dev).reshape(1, -1) for m in M]
        Xd = [self.X[m].dev("mean", self.dev) for m in M]

        def combine(views, cov, base, out):
            # out = lr * (sum_m tau_m (X_m - b_m) W_m) cov + (1 - lr) * base
            c = None
            Ps = []
            for m in views:
                P = fa.small_gemm(Wd[m], cov, alpha=taus[m])                        # tau_m W_m Sigma  [d_m x K]
                Ps.append(P)
                c = fa.small_gemm(bd[m], P, alpha=-1.0, beta=0.0 if c is None else 1.0, out=c)
            for i, m in enumerate(views):
                if i == 0:
                    fa.rows_gemm(Xd[m], Ps[i], c.reshape(-1), lr, base, 1.0 - lr, out)
                else:
                    fa.rows_gemm(Xd[m], Ps[i], None, lr, None, 0.0, out, accumulate=True)
            return out

        Zold = q.Z.dev("mean", self.dev)
        Znew = combine(list(M), Zcov, Zold, torch.empty_like(Zold))
        q.Z.set_dev("mean", Znew)
        zz = []
        if self.VARIANT == "A":
            for m in M:
                others = [j for j in M if j != m]
                if not others:
                    raise ValueError("variant A needs at least two views (reference :996 stacks the other views)")
                zp = combine(others, covs_pred[m], Znew, torch.empty_like(Znew))                     # :996-997
                q.Z.set_dev("mean_pred%d" % m, zp)
                zz.append(fa.atb(zp, zp, False, self.sc))
        ztz = self._refresh_zstats(also=zz)               # Z^T Z and the predictive Z^T Z of every view: one all-reduce
        q.Z.set_dev("prodT", fa.axpby(1.0, ztz, float(self.n_max), Zcov))                            # :989
        if self.VARIANT == "A":
            q.Z.set_dev("prodT_pred", [fa.axpby(1.0, zz[m], float(self.n_max), covs_pred[m]) for m in M])
            q.Z["mean_pred"] = _PredList(q.Z, self.m)

    def update_xs(self, m):
        """X* of a semi-supervised 'reg' view (:1157-1174) and its write-back into X (:778-783): the unobserved entries take
        Z W^T + b as mean and 1/tau as variance; the statistics of X are refreshed for update_tau / update_b / update_w."""
        q = self.q_dist
        X, XS = self.X[m], q.XS[m]
        tau = self._tau(m)
        xs = torch.empty(self.n_loc, self.d[m], dtype=F64, device=self.dev)
        fa.rows_gemm(q.Z.dev("mean", self.dev), self._wt(m), q.b[m].dev("mean", self.dev), 1.0, None, 0.0, xs)
        XS.set_dev("mean", xs)
        dict.__setitem__(XS, "cov", (1.0 / tau) * np.eye(self.d[m]))
        mask = self._ss_mask_dev[m]
        X.set_dev("mean", torch.where(mask, xs, X.dev("mean", self.dev)))                            # masked copy (data movement)
        X.set_dev("cov", X.dev("cov", self.dev).masked_fill(mask, 1.0 / tau))
        self._refresh_xstats(m)
        dict.__setitem__(X, "sumlogdet", dist.sum_scalar(self._ss_count[m], self.dev) * math.log(1.0 / tau))   # :781-783

    def _wt(self, m):
        """W_m.mean^T [K, d] on the device (the row GEMMs take their right operand row-major)."""
        return self.q_dist.W[m].dev("mean", self.dev).t().contiguous()

    def update_b(self, m, n=None):
        """:1077-1092 (n: the view's own row count for the initial b, :1986).  The two scalars (Sigma_b[0, 0], <b b^T>) stay on the
        device -- update_tau reads <b b^T> from there -- and reach the host with the next _sync_scalars."""
        q = self.q_dist
        bmean = torch.empty(1, self.d[m], dtype=F64, device=self.dev)
        scal = self._scal_of(m)
        fa.b_update(self._xsum[m], self._zsum, q.W[m].dev("mean", self.dev), self._tau(m), self.n_max if n is None else n, bmean, scal)
        q.b[m].set_dev("mean", bmean)
        self._pending_scal[m] = scal

    def _scal_of(self, m):
        """Device scalars of view m: [Sigma_b[0,0], <b b^T>, tau.a, tau.b]."""
        if "_scal_m" not in self.__dict__:
            self._scal_m = {}
        if m not in self._scal_m:
            self._scal_m[m] = torch.zeros(4, dtype=F64, device=self.dev)
        return self._scal_m[m]

    def _sync_scalars(self):
        """ONE read-back of what the updates left on the device for the host side of the coordinator: the b scalars of every
        view and the two sums of every multilabel update (entropy and bound term, :1649, :1678)."""
        ms, ks = sorted(self._pending_scal), sorted(self._pending_mult)
        if not ms and not ks:
            return
        flat = torch.cat([self._pending_scal[m][:2] for m in ms] + [self._pending_mult[k] for k in ks]).cpu().numpy()
        q = self.q_dist
        for i, m in enumerate(ms):
            bc, bp = flat[2 * i], flat[2 * i + 1]
            dict.__setitem__(q.b[m], "cov", bc * np.eye(self.d[m]))
            dict.__setitem__(q.b[m], "prodT", float(bp))
        o = 2 * len(ms)
        self._mult_elbo = getattr(self, "_mult_elbo", {})
        for i, k in enumerate(ks):
            dict.__setitem__(self.X[k], "sumlogdet", float(flat[o + 2 * i]))
            self._mult_elbo[k] = float(flat[o + 2 * i + 1])
        self._pending_scal, self._pending_mult = {}, {}

    def update_alpha(self, m):
        """:1094-1112"""
        q = self.q_dist
        ab = torch.empty(q.Kc, dtype=F64, device=self.dev)
        fa.alpha_update(q.W[m].dev("prodT", self.dev), self.hyper.alpha_b[m], self.d[m], ab)
        dict.__setitem__(q.alpha[m], "a", (self.hyper.alpha_a[m] + 0.5 * self.d[m]) / (self.d[m]))
        q.alpha[m].set_dev("b", ab)

    def update_tau(self, m):
        """:1121-1139; tr<X^T X> and <b b^T> are read on the device; tau itself comes back (the next kernels take it as an argument)"""
        q = self.q_dist
        xtz1 = dist.sum_(fa.atb(self.X[m].dev("mean", self.dev), q.Z.dev("mean", self.dev), True, self.sc))
        scal = self._scal_of(m)
        pend = m in self._pending_scal
        fa.tau_update(self._xsum[m], self._zsum, xtz1, 0.0, q.Z.dev("prodT", self.dev), q.W[m].dev("prodT", self.dev),
                      q.W[m].dev("mean", self.dev), q.b[m].dev("mean", self.dev), 0.0 if pend else float(q.b[m]["prodT"]),
                      self.hyper.tau_a[m], self.hyper.tau_b[m], self.n_max, scal, bprodT_dev=scal[1:2] if pend else None,
                      xtx1_dev=self._xtx1[m], covsum_dev=self._covsum[m])
        sc = scal.cpu().numpy()
        q.tau[m]["a"] = float(sc[2])
        q.tau[m]["b"] = np.array([sc[3]])

    def update_x(self, m):
        """multilabel update_x + update_xi (:1197-1229), one fused pass; its two sums stay on the device until _sync_scalars"""
        q = self.q_dist
        X = self.X[m]
        n, d = self.n_loc, self.d[m]
        zwb = torch.empty(n, d, dtype=F64, device=self.dev)
        fa.rows_gemm(q.Z.dev("mean", self.dev), self._wt(m), q.b[m].dev("mean", self.dev), 1.0, None, 0.0, zwb)
        mean = X.dev("mean", self.dev)
        cov = X.dev("cov", self.dev)
        self._pending_mult[m] = fa.mult_update(self.t[m].dev("mean", self.dev), zwb, q.xi[m], mean, cov, self._tau(m), self.sc, sync=False)
        X.set_dev("mean", mean)
        X.set_dev("cov", cov)

    def update_t(self, m):
        """t* of a semi-supervised multilabel view (:1176-1195) and t.mean[mask] = tS.mean[mask] (:861), one pass"""
        self._sync_scalars()
        t = self.t[m]
        tm = t.dev("mean", self.dev)
        sums = fa.mult_ss_update(self.X[m].dev("mean", self.dev), self._ss_mask_dev[m], tm, self.sc)
        t.set_dev("mean", tm)
        self.q_dist.tS[m]["sumlogdet"] = float(sums[0])
        self._mult_elbo[m] += float(sums[1])      # the bound term sum X.mean * t.mean (:1678) with the new labels

    def update_xi(self, m):
        """fused into update_x (the kernel writes xi = sqrt(cov + mean^2) in the same pass, :1229)"""

    # ---- lower bound (:1572-1688) --------------------------------------------------------------------------------
    def update_bound(self):
        q = self.q_dist
        self._sync_scalars()
        # slogdet of Sigma_Z and every Sigma_W (de-pruned: padded with 1e-15, :1603 on :518-537): the LU launches are queued
        # together and read back once
        covs = [q.Z.dev("cov", self.dev)] + [q.W[m].dev("cov", self.dev) for m in range(self.m)]
        ld = torch.empty(len(covs), 2, dtype=F64, device=self.dev)
        for i, c in enumerate(covs):
            fa.slogdet_into(c, ld[i])
        ld = ld.cpu().numpy()

        def hgauss(rows, i, prev):
            sgn, lad = ld[i]
            H = 0.5 * rows * (lad if sgn != 0 else -np.inf)
            return prev if abs(H) == np.inf else H

        q.Z["LH"] = hgauss(self.n_max, 0, q.Z["LH"])
        ent = q.Z["LH"]
        for m in range(self.m):
            q.W[m]["LH"] = hgauss(self.d[m], 1 + m, q.W[m]["LH"])
            bc = q.b[m]["cov"][0, 0]
            Hb = 0.5 * 1 * self.d[m] * math.log(bc) if bc > 0 else -np.inf                            # slogdet(bc * I_d)
            q.b[m]["LH"] = q.b[m]["LH"] if abs(Hb) == np.inf else Hb
            q.alpha[m]["LH"] = np.sum(-np.log(q.alpha[m]["b"] + EPS))
            q.tau[m]["LH"] = np.sum(-np.log(q.tau[m]["b"] + EPS))
            ent = ent + q.W[m]["LH"] + q.b[m]["LH"] + q.alpha[m]["LH"] + q.tau[m]["LH"]
            if self.method[m] == "mult":
                H = 0.5 * self.X[m]["sumlogdet"]
                dict.__setitem__(self.X[m], "LH", self.X[m]["LH"] if abs(H) == np.inf else H)
                ent = ent + self.X[m]["LH"]
            if self.SS[m] and self.method[m] == "reg":                                               # :1651-1653, 1666-1668
                H = 0.5 * self.X[m]["sumlogdet"]
                dict.__setitem__(q.XS[m], "LH", q.XS[m]["LH"] if abs(H) == np.inf else H)
                ent = ent + q.XS[m]["LH"]
            if self.SS[m] and self.method[m] == "mult":                                              # :1655-1656, 1669-1670
                H = 0.5 * q.tS[m]["sumlogdet"]
                q.tS[m]["LH"] = q.tS[m]["LH"] if abs(H) == np.inf else H
                ent = ent + q.tS[m]["LH"]
        q.Z["Elogp"] = -0.5 * np.trace(q.Z["prodT"])
        elogp = q.Z["Elogp"]
        for m in range(self.m):
            q.b[m]["Elogp"] = -0.5 * q.b[m]["prodT"]
            e_tau = -(0.5 * self.n_max * self.d[m] + self.hyper.tau_a[m] - 1) * np.log(q.tau[m]["b"] + EPS)
            if self.method[m] == "mult":
                e_tau = e_tau + self._mult_elbo[m]                                                    # :1678
            q.tau[m]["ElogpXtau"] = e_tau
            q.alpha[m]["ElogpWalp"] = -(0.5 * self.d[m] + self.hyper.alpha_a[m] - 1) * np.sum(np.log(q.alpha[m]["b"]))
            elogp = elogp + e_tau + q.alpha[m]["ElogpWalp"] + q.b[m]["Elogp"]
        return float(np.ravel(elogp - ent)[0])

    # ---- pruning (:490-537) ----------------------------------------------------------------------------------------
    def pruning(self, pruning_crit):
        q = self.q_dist
        keep = np.zeros(q.Kc, dtype=bool)
        for m in range(self.m):
            keep |= np.any(np.abs(q.W[m]["mean"][:, :q.Kc]) > pruning_crit, axis=0)
        sel = np.flatnonzero(keep)
        if len(sel) == q.Kc and not getattr(self, "_padded", False):
            return                     # every factor stays and nothing was padded: the device-resident state is left alone
        self._padded = False
        if len(sel) != q.Z.dev("mean", self.dev).shape[1]:
            if len(sel):
                Zd = q.Z.dev("mean", self.dev)
                out = torch.empty(self.n_loc, len(sel), dtype=F64, device=self.dev)
                fa.gather_cols(Zd, torch.from_numpy(sel.astype(np.int32)).to(self.dev), out)
                q.Z.set_dev("mean", out)
            else:
                q.Z["mean"] = np.zeros((self.n_loc, 0))
        q.Z["cov"] = q.Z["cov"][sel, :][:, sel]
        q.Z["prodT"] = q.Z["prodT"][sel, :][:, sel]
        for m in range(self.m):
            W = q.W[m]
            W["mean"] = W["mean"][:, sel]
            W["cov"] = W["cov"][sel, :][:, sel]
            W["prodT"] = W["prodT"][sel, :][:, sel]
            q.alpha[m]["b"] = q.alpha[m]["b"][sel]
        changed = len(sel) != q.Kc
        q.Kc = len(sel)
        if changed and q.Kc > 0:
            self._refresh_zstats()

    def depruning(self, pruning_crit):
        """Pads the K x K / d x K state back to the original Kc with 1e-15 before the bound (:518-537).  Z.mean is left
        at its pruned width on the device: the bound only reads its row count (:1603)."""
        q = self.q_dist
        Kp = self.Kc - q.Kc
        if Kp == 0:
            return
        self._padded = True

        def pad2(A):
            top = np.hstack((A, pruning_crit * np.ones((q.Kc, Kp))))
            return np.vstack((top, pruning_crit * np.ones((Kp, self.Kc))))

        q.Z["cov"], q.Z["prodT"] = pad2(q.Z["cov"]), pad2(q.Z["prodT"])
        for m in range(self.m):
            W = q.W[m]
            W["mean"] = np.hstack((W["mean"], pruning_crit * np.ones((self.d[m], Kp))))
            W["cov"], W["prodT"] = pad2(W["cov"]), pad2(W["prodT"])
            q.alpha[m]["b"] = np.hstack((q.alpha[m]["b"], pruning_crit * np.ones((Kp,))))

    # ---- pickling ----------------------------------------------------------------------------------------------------
    def __getstate__(self):
        if "_pending_scal" in self.__dict__ and "dev" in self.__dict__:
            self._sync_scalars()
            self._flush_info()
        st = self.__dict__.copy()
        for k in ("sc", "scal", "_zsum", "_xsum", "dev", "_ss_mask_dev", "_xtx1", "_covsum", "_scal_m", "_pending_info", "_pending_scal",
                  "_pending_mult", "_t1_stats"):
            st.pop(k, None)
        if "q_dist" in st:
            xi = st["q_dist"].xi
            st["q_dist"] = copy.copy(st["q_dist"])
            st["q_dist"].xi = [x.cpu().numpy() if isinstance(x, torch.Tensor) else x for x in xi]
        return st


class _PredList(object):
    """q.Z['mean_pred'][m] of variant A (:992-997): device-resident, materialised on indexing."""

    def __init__(self, zdict, m):
        self._z, self._m = zdict, m

    def __len__(self):
        return self._m

    def __getitem__(self, m):
        return self._z._dev["mean_pred%d" % m].cpu().numpy()
