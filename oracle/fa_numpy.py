"""NumPy fp64 restatement of the SSHIBA mean-field half of FA-VAE.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  Citations are path:line under /root/reference; "cA" = condition_a_VAE,
"DA" = domain_adaptation.

Only the branches the four reference trainers exercise are restated ('reg', 'mult', 'img' views; no sparse,
kernel or categorical paths -- SURVEY.md section 2 row 13; semi-supervised 'reg' and 'mult' views with SS_sep=0 are in).
"""
import math
import sys

import numpy as np
import scipy.linalg

EPS = sys.float_info.epsilon


# --------------------------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------------------------
def my_inverse(A):
    """cA/favae.py:904-922 -- inverse through the Cholesky factor; any failure gives NaN (caller skips)."""
    try:
        Li = scipy.linalg.pinv(np.linalg.cholesky(A), atol=1e-10, rtol=0.0)  # scipy 1.7.3 `rcond=1e-10`
        return Li.T @ Li
    except Exception:
        return np.nan


def sigmoid(x):
    """cA/favae.py:924-935."""
    return np.exp(-np.log(1 + np.exp(-x)))


def lambda_func(x):
    """cA/favae.py:937-949."""
    return np.exp(np.log(sigmoid(x) - 0.5) - np.log(2 * x))


def r2_score(y, yhat):
    """sklearn.metrics.r2_score (uniform average over outputs), used as a diagnostic cA/favae.py:840,848."""
    y = np.asarray(y, dtype=np.float64)
    yhat = np.asarray(yhat, dtype=np.float64)
    if y.ndim == 1:
        y, yhat = y[:, None], yhat[:, None]
    num = ((y - yhat) ** 2).sum(axis=0)
    den = ((y - y.mean(axis=0)) ** 2).sum(axis=0)
    nz = den != 0
    out = np.ones(y.shape[1])
    out[nz] = 1 - num[nz] / den[nz]
    out[(~nz) & (num != 0)] = 0.0
    return float(out.mean())


class Hyper:
    """cA/favae.py:1856-1881."""

    def __init__(self, m):
        self.alpha_a = [1e-14] * m
        self.alpha_b = [1e-14] * m
        self.tau_a = [1e-14] * m
        self.tau_b = [1e-14] * m


class QDist:
    """cA/favae.py:1883-2051, the parts reached with sparse=0, no *_init; SS = semi-supervised 'reg' views (:1963-1970)."""

    def __init__(self, X, n, n_max, d, Kc, m, method, hyper, SS=None):
        self.n, self.n_max, self.d, self.Kc, self.m, self.X = n, n_max, d, Kc, m, X
        self.SS = SS if SS is not None else [False] * m
        self.alpha = [dict(a=hyper.alpha_a[i], b=hyper.alpha_b[i] * np.ones(Kc), LH=None, ElogpWalp=None)
                      for i in range(m)]
        self.tau = [dict(a=hyper.tau_a[i], b=hyper.tau_b[i] * np.ones(1), LH=None, ElogpXtau=None)
                    for i in range(m)]
        self.xi = [np.sqrt(X[i]["cov"] + X[i]["mean"] ** 2) for i in range(m)]            # :1912-1914
        blank = lambda: dict(mean=None, cov=None, prodT=None, LH=0, Elogp=0)
        self.b = [blank() for _ in range(m)]
        self.W = [blank() for _ in range(m)]
        self.XS = [blank() for _ in range(m)]
        self.tS = [blank() for _ in range(m)]
        self.tc = {}
        Z = blank()
        Z["mean"] = np.random.normal(0.0, 1.0, n_max * Kc).reshape(n_max, Kc)            # :1948
        Z["cov"] = np.eye(Kc)
        Z["prodT"] = Z["cov"] + n_max * Z["cov"]                                        # :1950
        self.Z = Z
        for i in range(m):
            if self.SS[i]:                                                               # :1963-1970 (same draws, same order)
                self.tc[i] = np.random.rand(n_max, d[i])
                self.tS[i]["mean"] = np.random.randint(2, size=[n_max, d[i]])
                self.tS[i]["cov"] = np.eye(d[i])
                self.tS[i]["sumlogdet"] = 0
                self.XS[i]["mean"] = np.random.normal(0.0, 1.0, n_max * d[i]).reshape(n_max, d[i])
                self.XS[i]["cov"] = np.zeros(self.XS[i]["mean"].shape)
            W = self.W[i]
            scale = 1 / np.repeat(self.alpha_mean(i).reshape(1, Kc), d[i], axis=0)
            W["mean"] = np.random.normal(np.zeros((d[i], Kc)), scale)                    # :1977
            W["cov"] = W["mean"].T @ W["mean"]
            W["prodT"] = W["mean"].T @ W["mean"] + Kc * W["cov"]                        # :1979
            b = self.b[i]
            b["cov"] = (1 + n[i] * self.tau_mean(i)) ** (-1) * np.eye(d[i])              # :1986
            b["mean"] = self.tau_mean(i) * np.dot(
                np.sum(X[i]["mean"] - Z["mean"] @ W["mean"].T, axis=0)[np.newaxis, :], b["cov"])
            b["prodT"] = np.sum(b["mean"] ** 2) + d[i] * b["cov"][0, 0]                  # :1988

    def alpha_mean(self, m):
        return self.alpha[m]["a"] / self.alpha[m]["b"]

    def tau_mean(self, m):
        return self.tau[m]["a"] / self.tau[m]["b"]


def struct_data(X, method, latent_dim=None, lr=None, epochs=None, dataset=None):
    """cA/favae.py:585-621 (only the keys the restated paths read)."""
    return dict(data=X, method=method, latent_dim=latent_dim, lr=lr, epochs=epochs, dataset=dataset)


# --------------------------------------------------------------------------------------------------------
# coordinator
# --------------------------------------------------------------------------------------------------------
class SSHIBAOracle:
    """Restates SSHIBA.fit / fit_iterate / update / update_bound / pruning for variants
    'A' (cA/favae.py: predictive Z for the VAE prior, moment prior, schedule 50/5/5) and
    'B' (DA/favae.py: joint Z, schedule 50/20/20 with beta 0.1/0.5/1, sampled prior) and 'C' (TL/favae.py with
    betaVAE.py: the coordinator of 'B', but `favae` is never passed, so the VAE trains against N(0,I)).
    `vae_factory(dimx, channels, lr, dataset, latentspace_lr)` builds the image-view model
    (oracle.vae_torch.OracleImgVAE in the tests)."""

    def __init__(self, Kc=2, prune=0, latentspace_lr=1, variant="A", vae_factory=None, diagnostics=True):
        self.Kc, self.prune, self.latentspace_lr = int(Kc), int(prune), latentspace_lr
        self.variant, self.vae_factory, self.diagnostics = variant, vae_factory, diagnostics

    # -- fit: cA/favae.py:84-379 ---------------------------------------------------------------------------
    def fit(self, *views, max_iter=1, pruning_crit=1e-6, iterate=True):
        import copy
        self.n = [int(v["data"].shape[0]) for v in views]
        self.n_max = int(np.max(self.n))
        self.m = len(views)
        self.d, self.method, self.X, self.t = [], [], [], {}
        self.SS, self.SS_mask = [], {}
        self.img_vae = [None] * self.m
        self.keep_training = {}
        self.total_elbo, self.kl_q_p, self.recon_loss = [], [], []
        for m, v in enumerate(views):
            meth = v["method"]
            self.method.append(meth)
            if meth == "reg":
                self.d.append(int(v["data"].shape[1]))
            elif meth == "img":
                self.d.append(int(v["latent_dim"]))
            elif meth == "mult":
                if v["data"].ndim < 2:
                    v["data"] = v["data"][:, None]
                self.d.append(int(v["data"].shape[1]))
            else:
                raise ValueError(meth)
            d = self.d[m]
            mn = np.random.normal(0.0, 1.0, self.n_max * d).reshape(self.n_max, d)       # :179
            info = dict(data=np.random.normal(0.0, 1.0, self.n_max * d).reshape(self.n_max, d),
                        mean=mn, cov=mn ** 2, prodT=mn.T @ mn, LH=0, Elogp=0, sumlogdet=0, epochs=20)
            self.X.append(info)
            data = copy.deepcopy(v["data"])
            if meth == "reg":                                                            # :236-257
                info["data"] = data
                info["mean"] = np.copy(data)
                if self.n_max > self.n[m]:                                               # fewer rows: the rest is unobserved
                    info["mean"] = np.vstack((info["mean"], np.nan * np.ones((self.n_max - self.n[m], d))))
                mask = np.isnan(info["mean"])
                self.SS_mask[m] = mask
                if np.sum(mask) > 0:                                                     # semi-supervised view
                    self.SS.append(True)
                    for dd in range(d):
                        cnt = int(np.sum(mask[:, dd]))
                        if cnt == mask.shape[0]:
                            info["mean"][mask[:, dd], dd] = np.random.normal(0.0, 1.0, cnt)
                        else:
                            info["mean"][mask[:, dd], dd] = np.random.normal(np.nanmean(info["mean"][:, dd], axis=0),
                                                                             np.nanstd(info["mean"][:, dd], axis=0), cnt)
                    info["cov"][~mask] = 0.0                                             # the rest keeps its random init (:179)
                else:
                    self.SS.append(False)
                    info["cov"] = np.zeros(info["mean"].shape)
            elif meth == "img":                                                          # :260-267
                self.SS.append(False)
                self.keep_training[m] = True
                self.t[m] = copy.deepcopy(info)
                self.t[m]["data"] = data
                self.img_vae[m] = self.vae_factory(dimx=v["latent_dim"], channels=data.shape[1], lr=v["lr"],
                                                   dataset=v["dataset"], latentspace_lr=self.latentspace_lr)
            elif meth == "mult":                                                         # :269-346
                self.t[m] = copy.deepcopy(info)
                tdata = data.astype(float)
                if self.n_max > self.n[m]:
                    tdata = np.vstack((tdata, np.nan * np.ones((self.n_max - self.n[m], d))))
                self.t[m]["data"] = tdata
                mask = np.isnan(tdata)
                self.SS_mask[m] = mask
                tt = np.abs(tdata[~mask] - 0.05)
                info["mean"][~mask] = np.log(tt / (1 - tt))                              # :276
                info["cov"][~mask] = 0
                if np.sum(mask) > 0:                                                     # semi-supervised labels (SS_sep = 0)
                    self.SS.append(True)
                    for dd in range(d):                                                  # :278-280
                        obs, cnt = info["mean"][~mask[:, dd], dd], int(np.sum(mask[:, dd]))
                        info["mean"][mask[:, dd], dd] = np.random.normal(np.nanmean(obs, axis=0), np.nanstd(obs, axis=0), cnt)
                        info["cov"][mask[:, dd], dd] = np.abs(np.random.normal(np.nanmean(obs, axis=0), np.nanstd(obs, axis=0), cnt))
                    tm = self.t[m]["mean"]
                    for dd in range(d):                                                  # :285-286
                        tm[mask[:, dd], dd] = np.random.normal(np.nanmean(tm[:, dd], axis=0), np.nanstd(tm[:, dd], axis=0),
                                                               int(np.sum(mask[:, dd])))
                    tm[~mask] = tdata[~mask]
                    self.t[m]["cov"][~mask] = 0
                else:
                    self.SS.append(False)
                    self.t[m]["mean"] = np.copy(tdata)
                    self.t[m]["cov"] = np.zeros(tdata.shape)
        self.L = []
        self.metrics = {}
        self.hyper = Hyper(self.m)
        self.q_dist = QDist(self.X, self.n, self.n_max, self.d, self.Kc, self.m, self.method, self.hyper, self.SS)
        if iterate:
            self.fit_iterate(max_iter=max_iter, pruning_crit=pruning_crit)

    # -- fit_iterate: cA/favae.py:383-451 ------------------------------------------------------------------
    def fit_iterate(self, max_iter=1, pruning_crit=1e-6, tol=1e-5):
        q = self.q_dist
        for i in range(max_iter):
            self.update(iter_count=i)
            if self.prune:
                self.depruning(1e-15)
                self.L.append(self.update_bound())
                self.pruning(pruning_crit)
                if q.Kc == 0:
                    return
            else:
                self.L.append(self.update_bound())
            if len(self.L) > 100 and abs(1 - np.mean(self.L[-101:-1]) / self.L[-1]) < tol:
                return

    # -- update: cA/favae.py:743-902 ; DA/favae.py:759-921 -------------------------------------------------
    def update(self, iter_count=0):
        q = self.q_dist
        for m in range(self.m):
            self.update_w(m)
        self.update_Z()
        for m in range(self.m):
            meth = self.method[m]
            if meth == "reg":
                if self.SS[m]:                                                           # :775-783
                    X, mask = self.X[m], self.SS_mask[m]
                    self.update_xs(m)
                    X["mean"][mask] = q.XS[m]["mean"][mask]
                    X["cov"][mask] = q.XS[m]["cov"][0, 0]
                    X["prodT"] = X["mean"].T @ X["mean"] + np.diag(np.sum(X["cov"], axis=0))
                    X_SS = np.copy(X["cov"])
                    X_SS[~mask] = 1
                    X["sumlogdet"] = np.sum(np.log(X_SS))
                else:
                    self.X[m]["prodT"] = self.X[m]["mean"].T @ self.X[m]["mean"]       # :785
                self.update_tau(m)
                self.update_b(m)
            if meth == "img":
                self._update_img(m, iter_count)
            elif meth == "mult":
                for _ in range(2):                                                       # :855-861
                    self.update_x(m)
                    self.update_xi(m)
                    if self.SS[m]:
                        self.update_t(m)
                        self.t[m]["mean"][self.SS_mask[m]] = q.tS[m]["mean"][self.SS_mask[m]]
                self.update_tau(m)
            self.update_b(m)                                                             # :867-868
            self.update_alpha(m)                                                         # :872

    def _update_img(self, m, iter_count):
        q = self.q_dist
        vae = self.img_vae[m]
        if self.variant == "A":                                                          # cA :792-802
            if iter_count == 0:
                epochs, beta = 50, 1
            elif iter_count < 3:
                epochs, beta = 5, 1
            else:
                self.keep_training[m] = False
                epochs, beta = 5, 5
            Z_in = {"mean": q.Z["mean_pred"][m], "cov": q.Z["covs_pred"][m]}
        else:                                                                            # DA :806-812
            if iter_count == 0:
                epochs, beta = 50, 0.1
            elif iter_count == 1:
                epochs, beta = 20, 0.5
            else:
                self.keep_training[m] = False
                epochs, beta = 20, 1
            Z_in = q.Z
        if hasattr(self, "epochs_override") and self.epochs_override is not None:
            epochs = self.epochs_override[min(iter_count, len(self.epochs_override) - 1)]
        trained = False
        if self.keep_training[m]:
            # variant C (TL/favae.py:812 with betaVAE.py) never passes `favae`: the VAE keeps its N(0,I) prior
            self.keep_training[m] = vae.trainloop(img=self.t[m]["data"], Z=Z_in, W=q.W[m], b=q.b[m],
                                                  tau=q.tau_mean(m), epochs=epochs, favae=self.variant != "C", beta=beta)
            trained = True
        elif iter_count % 10 == 0:
            if self.variant == "A":                                                      # cA :821 (N(0,I) prior)
                vae.trainloop(img=self.t[m]["data"], Z=Z_in, W=q.W[m], b=q.b[m], tau=q.tau_mean(m),
                              epochs=epochs)
            else:                                                                        # DA :827
                vae.trainloop(img=self.t[m]["data"], Z=Z_in, W=q.W[m], b=q.b[m], tau=q.tau_mean(m),
                              epochs=epochs, favae=self.variant != "C")
            trained = True
        if trained:
            self.total_elbo.append(vae.elbo_training)
            self.recon_loss.append(vae.reconstruc_during_training)
            self.kl_q_p.append(vae.KL_QandP)
            self.X[m]["mean"], self.X[m]["cov"] = vae.update_x(img=self.t[m]["data"])    # :817
            self.X[m]["prodT"] = self.X[m]["mean"].T @ self.X[m]["mean"] + np.diag(np.sum(self.X[m]["cov"], axis=0))
        self.update_tau(m)
        self.update_b(m)
        if self.diagnostics:
            rec = vae.reconstruction(self.X[m]["mean"], self.X[m]["cov"])                # :839 (1 np.random.randn)
            self.metrics["R2 reconstruction %d" % m] = r2_score(self.t[m]["data"].ravel(), rec.ravel())
            if self.variant == "A" and self.m > 1 and self.method[1] == "mult":          # :846-848
                pred = self.predict_zero_input(1)
                self.metrics["r2_acc %d" % m] = r2_score(self.t[1]["data"], pred)

    def predict_zero_input(self, m_out):
        """cA/favae.py:846-848 calls predict([0],[1], struct_data(.., 'vae')): a view of method 'vae' adds
        nothing to Z_mean (:1402-1414), so the prediction is sigmoid(b * (1+pi/8 var)^-1/2) (:1469-1477)."""
        q = self.q_dist
        aux = np.eye(q.Kc) + q.tau_mean(0) * (q.W[0]["mean"].T @ q.W[0]["mean"])          # :1370-1373
        Z_cov = my_inverse(aux)
        Zm = np.zeros((self.n_max, q.Kc)) @ Z_cov
        m_x = Zm @ q.W[m_out]["mean"].T + q.b[m_out]["mean"]
        var_x = q.tau_mean(m_out) ** (-1) * np.eye(self.d[m_out]) + np.linalg.multi_dot(
            [q.W[m_out]["mean"], Z_cov, q.W[m_out]["mean"].T])
        mean_t = np.zeros((self.n_max, self.d[m_out]))
        for d in range(self.d[m_out]):
            mean_t[:, d] = sigmoid(m_x[:, d] * (1 + math.pi / 8 * var_x[d, d]) ** (-0.5))
        return mean_t

    # -- predict: cA/favae.py:1336-1483 (variant A), DA/favae.py:1345-1478 (variants B, C) ------------------
    def _predict_latent(self, m_in, args):
        """:1363-1423 -- posterior of Z given the input views.  Only dict inputs of method 'img' / 'mult' add to
        Z_mean in the reference (a list input fails at :1378, any other dict adds nothing)."""
        q = self.q_dist
        n_pred = np.shape(args[0]["data"])[0]
        aux = np.eye(q.Kc)
        for m in m_in:
            aux += q.tau_mean(m) * np.dot(q.W[m]["mean"].T, q.W[m]["mean"])
        Z_cov = my_inverse(aux)
        X_sample = None
        if not np.any(np.isnan(Z_cov)):
            self.Z_mean = np.zeros((n_pred, q.Kc))
            for m, arg in enumerate(args):
                mi = m_in[m]
                if arg["method"] == "img":
                    X_mean, X_cov = self.img_vae[mi].update_x(arg["data"])
                    X_sample = X_mean + np.random.randn() * np.sqrt(X_cov)
                    self.Z_mean += np.dot(X_sample - q.b[mi]["mean"], q.W[mi]["mean"]) * q.tau_mean(mi)
                if arg["method"] == "mult":
                    X_mean = np.log(np.abs((arg["data"] - 0.05)) / (1 - np.abs((arg["data"] - 0.05))))
                    self.Z_mean += np.dot(X_mean - q.b[mi]["mean"], q.W[mi]["mean"]) * q.tau_mean(mi)
            self.Z_mean = np.dot(self.Z_mean, Z_cov)
        return n_pred, Z_cov, X_sample

    def _predict_outputs(self, m_outs, n_pred, Z_cov, X_sample, sameimage):
        import torch
        q = self.q_dist
        predictions = {}
        if isinstance(m_outs, int):
            m_outs = [m_outs]
        for m_out in m_outs:
            W, b = q.W[m_out]["mean"], q.b[m_out]["mean"]
            if self.method[m_out] == "reg":
                mean_x = np.dot(self.Z_mean, W.T) + b
                var_x = q.tau_mean(m_out) ** (-1) * np.eye(self.d[m_out]) + np.linalg.multi_dot([W, Z_cov, W.T])
                predictions["output_view" + str(m_out)] = {"mean_x": mean_x, "var_x": var_x}
            if self.method[m_out] == "img":
                if self.variant == "A" and not sameimage:                                # cA :1438-1446
                    sampleZ = self.Z_mean + np.random.normal() * np.sqrt(np.diag(Z_cov))
                    sampleX = sampleZ @ W.T - b                                          # minus, as written
                    pred_img = self.img_vae[m_out].decode_eval(torch.tensor(sampleX).float())
                    predictions = pred_img, self.Z_mean, 0
                else:
                    Z_in = {"mean": self.Z_mean, "cov": Z_cov}
                    pred_img, xcart = self.img_vae[m_out].predict(Z=Z_in, W=q.W[m_out], b=q.b[m_out], tau=q.tau_mean(m_out))
                    if self.variant == "A":                                              # predict_sameimage :1544
                        predictions = pred_img, self.Z_mean, xcart
                    else:                                                                # DA :1450
                        if X_sample is None:
                            raise NameError("name 'X_sample' is not defined")          # what the reference does
                        predictions = pred_img, X_sample, Z_in, xcart
            elif self.method[m_out] == "mult":
                m_x = np.dot(self.Z_mean, W.T) + b
                var_x = q.tau_mean(m_out) ** (-1) * np.eye(self.d[m_out]) + np.linalg.multi_dot([W, Z_cov, W.T])
                mean_t = np.zeros((n_pred, self.d[m_out]))
                var_t = np.zeros((n_pred, self.d[m_out]))
                for d in np.arange(self.d[m_out]):
                    s = m_x[:, d] * (1 + math.pi / 8 * var_x[d, d]) ** (-0.5)
                    mean_t[:, d] = sigmoid(s)
                    var_t[:, d] = np.exp(s) / (1 + np.exp(s) ** 2)                       # :1479, as written
                predictions["output_view" + str(m_out)] = {"mean_x": mean_t, "var_x": var_t}
        return predictions

    def predict(self, m_in, m_outs, *args):
        n_pred, Z_cov, X_sample = self._predict_latent(m_in, args)
        return self._predict_outputs(m_outs, n_pred, Z_cov, X_sample, False)

    def predict_sameimage(self, m_in, m_outs, *args):
        """cA/favae.py:1485-1570 (variant A only)."""
        n_pred, Z_cov, X_sample = self._predict_latent(m_in, args)
        return self._predict_outputs(m_outs, n_pred, Z_cov, X_sample, True)

    # -- mean-field updates --------------------------------------------------------------------------------
    def update_w(self, m):
        """cA/favae.py:1058-1075 (non-sparse branch)."""
        q = self.q_dist
        tmp = 1 / np.sqrt(q.alpha_mean(m))
        aux = np.outer(tmp, tmp) * q.Z["prodT"] + np.eye(q.Kc) / q.tau_mean(m)
        cho = np.linalg.cholesky(aux)
        w_cov = 1 / q.tau_mean(m) * np.outer(tmp, tmp) * np.dot(scipy.linalg.pinv(cho.T), scipy.linalg.pinv(cho))
        if not np.any(np.isnan(w_cov)):
            W = q.W[m]
            W["cov"] = w_cov
            W["mean"] = q.tau_mean(m) * np.linalg.multi_dot(
                [np.subtract(self.X[m]["mean"], q.b[m]["mean"]).T, q.Z["mean"], W["cov"]])
            W["prodT"] = W["mean"].T @ W["mean"] + self.d[m] * W["cov"]

    def update_Z(self):
        """cA/favae.py:961-1000 (variant A, with the carried-over `aux`), DA/favae.py:980-1009 (variant B)."""
        q = self.q_dist
        M = range(self.m)
        aux = np.eye(q.Kc) + np.sum([q.tau_mean(m) * q.W[m]["prodT"] for m in M], axis=0)
        Z_cov = my_inverse(aux)
        if self.variant == "A":
            q.Z["covs_pred"] = []
            aux = np.eye(q.Kc)
            for m in M:
                others = sum(q.tau_mean(j) * q.W[j]["prodT"] for j in M if j != m)
                aux = aux + others                                                       # never reset (:979)
                q.Z["covs_pred"].append(my_inverse(aux))
        if np.any(np.isnan(Z_cov)):
            return
        q.Z["cov"] = Z_cov
        proj = [np.dot(np.subtract(self.X[m]["mean"], q.b[m]["mean"]), q.W[m]["mean"]) * q.tau_mean(m) for m in M]
        mn = np.stack(proj).sum(axis=0)
        lr = self.latentspace_lr
        q.Z["mean"] = lr * np.dot(mn, q.Z["cov"]) + (1 - lr) * q.Z["mean"]              # batchRate :952-959
        q.Z["prodT"] = q.Z["mean"].T @ q.Z["mean"] + self.n_max * q.Z["cov"]
        if self.variant == "A":
            q.Z["mean_pred"], q.Z["prodT_pred"] = [], []
            for m in M:
                mn = np.stack([proj[j] for j in M if j != m]).sum(axis=0)
                zp = lr * np.dot(mn, q.Z["covs_pred"][m]) + (1 - lr) * q.Z["mean"]       # :997
                q.Z["mean_pred"].append(zp)
                q.Z["prodT_pred"].append(zp.T @ zp + self.n_max * q.Z["covs_pred"][m])

    def update_b(self, m):
        """cA/favae.py:1077-1092."""
        q = self.q_dist
        b = q.b[m]
        b["cov"] = (1 + self.n_max * q.tau_mean(m)) ** (-1) * np.eye(self.d[m])
        resid = np.sum(np.subtract(self.X[m]["mean"], np.dot(q.Z["mean"], q.W[m]["mean"].T)), axis=0)
        b["mean"] = q.tau_mean(m) * np.dot(resid[np.newaxis, :], b["cov"])
        b["prodT"] = np.sum(b["mean"] ** 2) + self.d[m] * b["cov"][0, 0]

    def update_alpha(self, m):
        """cA/favae.py:1094-1112."""
        q = self.q_dist
        q.alpha[m]["a"] = (self.hyper.alpha_a[m] + 0.5 * self.d[m]) / self.d[m]
        q.alpha[m]["b"] = (self.hyper.alpha_b[m] + 0.5 * np.diag(q.W[m]["prodT"])) / self.d[m]

    def update_tau(self, m):
        """cA/favae.py:1121-1139."""
        q = self.q_dist
        dn = self.d[m] * self.n_max
        q.tau[m]["a"] = (self.hyper.tau_a[m] + 0.5 * dn) / dn
        X, W, Z, b = self.X[m], q.W[m], q.Z, q.b[m]
        s = (np.trace(X["prodT"]) + np.trace(np.dot(W["prodT"], Z["prodT"]))
             - 2 * np.trace(np.linalg.multi_dot([W["mean"], Z["mean"].T, X["mean"]]))
             + 2 * np.sum(np.linalg.multi_dot([Z["mean"], W["mean"].T, b["mean"].T]))
             - 2 * np.sum(np.dot(X["mean"], b["mean"].T))
             + self.n_max * b["prodT"])
        q.tau[m]["b"] = (self.hyper.tau_b[m] + 0.5 * s) / dn

    def update_x(self, m):
        """cA/favae.py:1197-1214 (multilabel X)."""
        q = self.q_dist
        X = self.X[m]
        X["cov"] = (q.tau_mean(m) + 2 * lambda_func(q.xi[m])) ** (-1)
        X["mean"] = (self.t[m]["mean"] - 0.5 + q.tau_mean(m) * (np.dot(q.Z["mean"], q.W[m]["mean"].T)
                                                                   + q.b[m]["mean"])) * X["cov"]
        X["prodT"] = X["mean"].T @ X["mean"] + np.diag(np.sum(X["cov"], axis=0))
        X["sumlogdet"] = np.sum(np.log(X["cov"]))

    def update_xs(self, m):
        """cA/favae.py:1157-1174 (X* of a semi-supervised 'reg' view)."""
        q = self.q_dist
        q.XS[m]["cov"] = (q.tau_mean(m) ** (-1) * np.eye(self.d[m])).astype(float)
        q.XS[m]["mean"] = np.add(np.dot(q.Z["mean"], q.W[m]["mean"].T), q.b[m]["mean"])

    def update_t(self, m):
        """cA/favae.py:1176-1195 (t* of a semi-supervised multilabel view)."""
        tS = self.q_dist.tS[m]
        tS["mean"] = sigmoid(self.X[m]["mean"])
        tS["cov"] = np.exp(np.subtract(np.log(tS["mean"]), np.log((1 + np.exp(self.X[m]["mean"])))))
        tS["sumlogdet"] = np.sum(np.log(tS["cov"]))

    def update_xi(self, m):
        """cA/favae.py:1216-1229."""
        self.q_dist.xi[m] = np.sqrt(self.X[m]["cov"] + self.X[m]["mean"] ** 2)

    # -- bound: cA/favae.py:1572-1688 ----------------------------------------------------------------------
    @staticmethod
    def _hgauss(mn, cov, prev):
        H = 0.5 * mn.shape[0] * np.linalg.slogdet(cov)[1]
        return prev if abs(H) == np.inf else H

    def update_bound(self):
        q = self.q_dist
        q.Z["LH"] = self._hgauss(q.Z["mean"], q.Z["cov"], q.Z["LH"])
        ent = q.Z["LH"]
        for m in range(self.m):
            q.W[m]["LH"] = self._hgauss(q.W[m]["mean"], q.W[m]["cov"], q.W[m]["LH"])
            q.b[m]["LH"] = self._hgauss(q.b[m]["mean"], q.b[m]["cov"], q.b[m]["LH"])
            q.alpha[m]["LH"] = np.sum(-np.log(q.alpha[m]["b"] + EPS))
            q.tau[m]["LH"] = np.sum(-np.log(q.tau[m]["b"] + EPS))
            ent = ent + q.W[m]["LH"] + q.b[m]["LH"] + q.alpha[m]["LH"] + q.tau[m]["LH"]
            if self.method[m] == "mult":
                H = 0.5 * self.X[m]["sumlogdet"]
                self.X[m]["LH"] = self.X[m]["LH"] if abs(H) == np.inf else H
                ent = ent + self.X[m]["LH"]
            if self.SS[m] and self.method[m] == "reg":                                   # :1651-1653, 1666-1668
                H = 0.5 * self.X[m]["sumlogdet"]
                q.XS[m]["LH"] = q.XS[m]["LH"] if abs(H) == np.inf else H
                ent = ent + q.XS[m]["LH"]
            if self.SS[m] and self.method[m] == "mult":                                  # :1655-1656, 1669-1670
                H = 0.5 * q.tS[m]["sumlogdet"]
                q.tS[m]["LH"] = q.tS[m]["LH"] if abs(H) == np.inf else H
                ent = ent + q.tS[m]["LH"]
        elogp = -0.5 * np.trace(q.Z["prodT"])
        for m in range(self.m):
            e_tau = -(0.5 * self.n_max * self.d[m] + self.hyper.tau_a[m] - 1) * np.log(q.tau[m]["b"] + EPS)
            if self.method[m] == "mult":
                e_tau = e_tau + np.sum(np.log(sigmoid(q.xi[m])) + self.X[m]["mean"] * self.t[m]["mean"]
                                       - 0.5 * (self.X[m]["mean"] + q.xi[m]))
            e_alpha = -(0.5 * self.d[m] + self.hyper.alpha_a[m] - 1) * np.sum(np.log(q.alpha[m]["b"]))
            elogp = elogp + e_tau + e_alpha - 0.5 * q.b[m]["prodT"]
        return (elogp - ent).item() if np.ndim(elogp - ent) else float(elogp - ent)

    # -- pruning: cA/favae.py:490-537 ----------------------------------------------------------------------
    def pruning(self, crit):
        q = self.q_dist
        keep = np.zeros(q.Kc, dtype=bool)
        for m in range(self.m):
            keep |= np.any(np.abs(q.W[m]["mean"][:, :q.Kc]) > crit, axis=0)   # only the first q.Kc (:502)
        sel = np.flatnonzero(keep)
        q.Z["mean"] = q.Z["mean"][:, sel]
        q.Z["cov"] = q.Z["cov"][sel, :][:, sel]
        q.Z["prodT"] = q.Z["prodT"][sel, :][:, sel]
        for m in range(self.m):
            W = q.W[m]
            W["mean"] = W["mean"][:, sel]
            W["cov"] = W["cov"][sel, :][:, sel]
            W["prodT"] = W["prodT"][sel, :][:, sel]
            q.alpha[m]["b"] = q.alpha[m]["b"][sel]
        q.Kc = len(sel)

    def depruning(self, val):
        q = self.q_dist
        Kp = self.Kc - q.Kc

        def pad2(A):
            top = np.hstack((A, val * np.ones((q.Kc, Kp))))
            return np.vstack((top, val * np.ones((Kp, self.Kc))))

        q.Z["mean"] = np.hstack((q.Z["mean"], val * np.ones((self.n_max, Kp))))
        q.Z["cov"], q.Z["prodT"] = pad2(q.Z["cov"]), pad2(q.Z["prodT"])
        for m in range(self.m):
            W = q.W[m]
            W["mean"] = np.hstack((W["mean"], val * np.ones((self.d[m], Kp))))
            W["cov"], W["prodT"] = pad2(W["cov"]), pad2(W["prodT"])
            q.alpha[m]["b"] = np.hstack((q.alpha[m]["b"], val * np.ones((Kp,))))
