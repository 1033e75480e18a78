"""torch-CPU functional restatement of the conv VAE half of FA-VAE.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).  Citations are path:line under /root/reference ("cA" = condition_a_VAE).

The network, loss, optimiser and bookkeeping follow cA/celeba_deepvanilla_vae.py; the `prior` argument of
OracleImgVAE selects the prior construction of the other plugin modules:
   'moment'  cA/celeba_deepvanilla_vae.py:197-210   (Z.mean @ W.mean^T + b.mean, beta forced to 1 :185)
   'sampled' domain_adaptation/celeba2cartoon_vae.py:140-155, transfer_learning/betaVAE.py:144-152,
             cA/celeba_bettervanilla_vae.py:145-150  (one scalar normal draw per factor matrix)
Arithmetic is torch's own (conv/BN/Adam/Normal-KL live in torch, which is present in the image); this file
restates the *calls, their order, dtypes and RNG consumption*.
"""
import copy
import math

import numpy as np
import torch
import torch.nn.functional as F

HID = (64, 128, 256, 512, 1024)          # cA/celeba_deepvanilla_vae.py:26
LRELU = 0.01                              # nn.LeakyReLU() default
BN_EPS, BN_MOM = 1e-5, 0.1                # nn.BatchNorm2d defaults


def _spec(zdim, channels):
    """(state_dict prefix, kind, shape args) in the construction order of cA/celeba_deepvanilla_vae.py:25-78;
    the order fixes the order in which the global torch generator is consumed by the initialisers."""
    s, c = [], channels
    for i, h in enumerate(HID):
        s += [("enc.%d.0" % i, "conv", (c, h)), ("enc.%d.1" % i, "bn", (h,))]
        c = h
    s += [("fc_mu", "lin", (HID[-1] * 4, zdim)), ("fc_var", "lin", (HID[-1] * 4, zdim)),
          ("decoder_input", "lin", (zdim, HID[-1] * 4))]
    r = HID[::-1]
    for i in range(4):
        s += [("dec.%d.0" % i, "convT", (r[i], r[i + 1])), ("dec.%d.1" % i, "bn", (r[i + 1],))]
    s += [("final_layer.0", "convT", (r[-1], r[-1])), ("final_layer.1", "bn", (r[-1],)),
          ("final_layer.3", "conv", (r[-1], channels))]
    return s


def init_state(zdim, channels):
    """Draw initial parameters exactly as constructing the reference nn.Module tree would (same module
    classes in the same order => same draws from the global generator).  Returns (params, buffers)."""
    P, B = {}, {}
    for name, kind, a in _spec(zdim, channels):
        if kind == "conv":
            mod = torch.nn.Conv2d(a[0], a[1], kernel_size=3, stride=2, padding=1)
        elif kind == "convT":
            mod = torch.nn.ConvTranspose2d(a[0], a[1], kernel_size=3, stride=2, padding=1, output_padding=1)
        elif kind == "lin":
            mod = torch.nn.Linear(a[0], a[1])
        else:
            mod = torch.nn.BatchNorm2d(a[0])
        for k, v in mod.named_parameters():
            P[name + "." + k] = v.detach().clone().requires_grad_(True)
        for k, v in mod.named_buffers():
            B[name + "." + k] = v.detach().clone()
    return P, B


def _block(x, P, B, pre, transposed, train):
    if transposed:
        x = F.conv_transpose2d(x, P[pre + ".0.weight"], P[pre + ".0.bias"], stride=2, padding=1, output_padding=1)
    else:
        x = F.conv2d(x, P[pre + ".0.weight"], P[pre + ".0.bias"], stride=2, padding=1)
    rm, rv = B[pre + ".1.running_mean"], B[pre + ".1.running_var"]
    if train:
        B[pre + ".1.num_batches_tracked"] += 1
    x = F.batch_norm(x, rm, rv, P[pre + ".1.weight"], P[pre + ".1.bias"], train, BN_MOM, BN_EPS)
    return F.leaky_relu(x, LRELU)


def encoder(x, P, B, train):
    """cA/celeba_deepvanilla_vae.py:80-87."""
    for i in range(5):
        x = _block(x, P, B, "enc.%d" % i, False, train)
    x = torch.flatten(x, start_dim=1)
    mu = F.linear(x, P["fc_mu.weight"], P["fc_mu.bias"])
    log_var = F.linear(x, P["fc_var.weight"], P["fc_var.bias"]) + 1e-6
    return mu, torch.exp(0.5 * log_var)


def decoder(z, P, B, train):
    """cA/celeba_deepvanilla_vae.py:94-99."""
    x = F.linear(z, P["decoder_input.weight"], P["decoder_input.bias"]).view(-1, 1024, 2, 2)
    for i in range(4):
        x = _block(x, P, B, "dec.%d" % i, True, train)
    # final_layer: ConvT, BN, LeakyReLU share the block form under the prefix 'final_layer' (.0/.1)
    x = F.conv_transpose2d(x, P["final_layer.0.weight"], P["final_layer.0.bias"], stride=2, padding=1, output_padding=1)
    if train:
        B["final_layer.1.num_batches_tracked"] += 1
    x = F.batch_norm(x, B["final_layer.1.running_mean"], B["final_layer.1.running_var"],
                     P["final_layer.1.weight"], P["final_layer.1.bias"], train, BN_MOM, BN_EPS)
    x = F.leaky_relu(x, LRELU)
    x = F.conv2d(x, P["final_layer.3.weight"], P["final_layer.3.bias"], padding=1)
    return torch.sigmoid(x)


def gaussian_ll(x, xr, var_x):
    """cA/celeba_deepvanilla_vae.py:152-171."""
    D = x.shape[1] * x.shape[2] * x.shape[3]
    x, xr = x.reshape(-1, D), xr.reshape(-1, D)
    v = torch.ones_like(xr) * var_x
    cnt = D * np.log(2 * np.pi) + torch.sum(torch.log(v), dim=-1)
    return (-0.5 * (cnt + torch.sum((x - xr) * v ** -1 * (x - xr), dim=-1))).mean()


def kl_div(mu_q, std_q, mu_p, std_p):
    """cA/celeba_deepvanilla_vae.py:137-150."""
    std_q = torch.clamp(std_q, 1e-6, 1e20)
    std_p = torch.clamp(std_p, 1e-6, 1e20)
    mu_q = torch.clamp(mu_q, -1e20, 1e20)
    mu_p = torch.clamp(mu_p, -1e20, 1e20)
    q = torch.distributions.Normal(mu_q, std_q)
    p = torch.distributions.Normal(mu_p, std_p)
    return torch.distributions.kl_divergence(q, p).sum(axis=1).mean()


def loss_terms(P, B, images, mu_P, tau, var_x, beta, eps=None):
    """One forward of cA/celeba_deepvanilla_vae.py:231-247.  `eps` None => drawn from the global generator
    exactly where reparameterize (:89-92) draws it."""
    std_P = torch.sqrt(torch.ones_like(mu_P) * (1 / tau))
    mu_Q, std_Q = encoder(images, P, B, True)
    if eps is None:
        eps = torch.randn_like(std_Q)
    z = eps.mul(std_Q).add_(mu_Q)
    rec = decoder(z, P, B, True)
    rec_err = -gaussian_ll(images, rec, var_x)
    kl = kl_div(mu_Q, std_Q, mu_P, std_P)
    loss = rec_err + beta * kl
    return loss, rec_err, kl, mu_Q, std_Q, z, rec


def draw_epoch_order(n, shuffle):
    """RNG consumption of one `for batch in DataLoader(...)` pass (torch/utils/data/dataloader.py
    _BaseDataLoaderIter.__init__ draws a base seed; RandomSampler.__iter__ draws the permutation seed)."""
    torch.empty((), dtype=torch.int64).random_()
    if not shuffle:
        return list(range(n))
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


class OracleImgVAE:
    """Mirror of ImgVAE (cA/celeba_deepvanilla_vae.py:108-430) on functional torch-CPU ops."""

    def __init__(self, dimx=2, channels=1, var_x=0.01, lr=1e-5, h=None, w=None, dataset=None, latentspace_lr=1,
                 prior="moment", force_beta_one=True, batch_size=64, update_x_self_img=False):
        self.dimx = self.latentdim = dimx
        self.channels, self.var_x, self.lr, self.dataset, self.latentspace_lr = channels, var_x, lr, dataset, latentspace_lr
        self.prior, self.force_beta_one, self.batch_size = prior, force_beta_one, batch_size
        self.update_x_self_img = update_x_self_img      # DA/celeba2cartoon_vae.py:240-258 encodes self.img whatever it is handed
        self.P, self.B = init_state(dimx, channels)
        self.names = list(self.P.keys())
        self.optim = torch.optim.Adam([self.P[k] for k in self.names], self.lr)
        self.img = None

    # -- helpers ---------------------------------------------------------------------------------------------
    def state_dict(self):
        sd = {k: v.detach().clone() for k, v in self.P.items()}
        sd.update({k: v.clone() for k, v in self.B.items()})
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k in self.P:
                self.P[k].copy_(sd[k])
            for k in self.B:
                self.B[k].copy_(sd[k])

    def prior_mean(self, Z, W, b):
        if self.prior == "moment":
            return torch.from_numpy(Z["mean"] @ W["mean"].T + b["mean"])
        Zs = Z["mean"] + np.random.normal() * np.sqrt(np.diag(Z["cov"]))
        Ws = W["mean"] + np.random.normal() * np.sqrt(np.diag(W["cov"]))
        bs = b["mean"] + np.random.normal() * np.sqrt(np.diag(b["cov"]))
        return torch.from_numpy(Zs @ Ws.T + bs)

    # -- trainloop: cA/celeba_deepvanilla_vae.py:173-295 -------------------------------------------------------
    def trainloop(self, img=None, Z=None, W=None, b=None, tau=None, epochs=20, wandb=None, favae=False, beta=1):
        if self.force_beta_one:
            beta = 1
        if img is not None:
            self.img = torch.tensor(img)
        self.loss_during_training, self.elbo_training = [], []
        self.reconstruc_during_training, self.KL_QandP = [], []
        if favae:
            prior_mean = self.prior_mean(Z, W, b)
            tau = torch.Tensor(np.asarray(tau))
        else:
            prior_mean = torch.zeros((img.shape[0], self.dimx))
            tau = torch.tensor(1)
        n = self.img.shape[0]
        nb = (n + self.batch_size - 1) // self.batch_size
        elbo_checker = -1e20
        for e in range(int(epochs)):
            tl = te = tr = tk = 0
            order = draw_epoch_order(n, True)
            for s in range(nb):
                idx = order[s * self.batch_size:(s + 1) * self.batch_size]
                images, mu_P = self.img[idx], prior_mean[idx]
                self.optim.zero_grad()
                loss, rec_err, kl, *_ = loss_terms(self.P, self.B, images, mu_P, tau, self.var_x, beta)
                loss.backward()
                self.optim.step()
                tl += loss.data.numpy()
                te += (-loss).data.numpy()
                tr += rec_err.data.numpy()
                tk += kl.data.numpy()
            self.loss_during_training.append(tl / nb)
            self.elbo_training.append(te / nb)
            self.reconstruc_during_training.append(tr / nb)
            self.KL_QandP.append(tk / nb)
            elbo_checker = self.restore_weights(elbo_checker, e)
        return abs(1 - np.mean(self.elbo_training[-10:]) / self.elbo_training[-1]) > 1e-4

    def restore_weights(self, elbo_checker, e):
        """cA/celeba_deepvanilla_vae.py:297-329; the file round trip is kept in memory."""
        if self.elbo_training[-1] > elbo_checker:
            self._ckp = dict(elbo=self.elbo_training[-1], state_dict=self.state_dict(),
                             optimizer=copy.deepcopy(self.optim.state_dict()))
            elbo_checker = self.elbo_training[-1]
        if self.elbo_training[-1] < elbo_checker:
            self.load_state_dict(self._ckp["state_dict"])
            self.optim.load_state_dict(copy.deepcopy(self._ckp["optimizer"]))
            elbo_checker = self._ckp["elbo"]
        return elbo_checker

    # -- update_x: cA/celeba_deepvanilla_vae.py:331-358 --------------------------------------------------------
    def update_x(self, img=None):
        n_out = img.shape[0] if img is not None else self.img.shape[0]
        img = torch.Tensor(img) if (img is not None and not self.update_x_self_img) else self.img
        n = img.shape[0]
        if n != n_out:
            raise ValueError("could not broadcast input array from shape (%d,%d) into shape (%d,%d)" % (n, self.latentdim, n_out, self.latentdim))
        mu, var = np.zeros((n, self.latentdim)), np.zeros((n, self.latentdim))
        draw_epoch_order(n, False)
        with torch.no_grad():
            for s in range(0, n, self.batch_size):
                m_, s_ = encoder(img[s:s + self.batch_size], self.P, self.B, False)
                mu[s:s + self.batch_size] = m_.numpy()
                var[s:s + self.batch_size] = s_.pow(2).numpy()
        return mu, var

    # -- reconstruction: cA/celeba_deepvanilla_vae.py:360-385 --------------------------------------------------
    def reconstruction(self, mean, var):
        sample = torch.Tensor(mean + np.random.randn() * np.sqrt(var))
        n = sample.shape[0]
        out = np.zeros((n,) + tuple(self.img.shape[1:]))
        draw_epoch_order(n, False)
        with torch.no_grad():
            for s in range(0, n, self.batch_size):
                out[s:s + self.batch_size] = decoder(sample[s:s + self.batch_size], self.P, self.B, False).numpy()
        return out

    # -- predict: cA/celeba_deepvanilla_vae.py:387-430 ----------------------------------------------------------
    def predict(self, Z=None, W=None, b=None, tau=None, mean_x=None, var_x=None):
        if mean_x is None:
            Zs = Z["mean"] + np.random.normal() * np.sqrt(np.diag(Z["cov"]))
            Ws = W["mean"] + np.random.normal() * np.sqrt(np.diag(W["cov"]))
            bs = b["mean"] + np.random.normal() * np.sqrt(np.diag(b["cov"]))
            mean = torch.from_numpy(Zs @ Ws.T + bs)
            std = torch.sqrt(torch.ones_like(mean) / torch.from_numpy(np.asarray(1 / tau)))
        else:
            mean, std = torch.from_numpy(mean_x), torch.sqrt(torch.from_numpy(var_x))
        rsample = torch.distributions.Normal(mean, std).rsample().float()
        n = rsample.shape[0]
        out = np.zeros((n,) + tuple(self.img.shape[1:]))
        draw_epoch_order(n, False)
        with torch.no_grad():
            for s in range(0, n, self.batch_size):
                out[s:s + self.batch_size] = decoder(rsample[s:s + self.batch_size], self.P, self.B, False).numpy()
        return out, rsample

    def decode_eval(self, z):
        """`self.decoder(z)` as SSHIBA.predict calls it (cA/favae.py:1445) on a module left in eval mode by the
        reconstruction diagnostic / update_x that precede it."""
        with torch.no_grad():
            return decoder(z, self.P, self.B, False)
