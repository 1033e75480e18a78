"""Host-side mirror of the reference VAE plugin modules (`class VAE(nn.Module)`, `class ImgVAE(VAE)`), backed by the
HIP executor.  Mirrors condition_a_VAE/celeba_deepvanilla_vae.py (and the three sibling plugins, which differ only
in how the FA prior is built and in the trainloop signature): same constructor, same method names / return types,
same attributes, same state_dict keys, same consumption of the global NumPy / torch generators.

The nn.Module tree only *holds* parameters (so that initialisation draws, state_dict(), parameters(), pickling and
torch.optim.Adam state look exactly like the reference's); every forward / backward / optimiser step runs in
libfavae_hip.so.  There is no CPU fallback: without a GPU the compute methods raise.
"""
import os

import numpy as np
import torch
from torch import nn

from . import dist
from . import lib as L

_WIDTHS = (64, 128, 256, 512, 1024)


def _local_device():
    if not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))


def _stack(kind, cin, cout):
    conv = (nn.Conv2d(cin, out_channels=cout, kernel_size=3, stride=2, padding=1) if kind == "down" else
            nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1))
    return nn.Sequential(conv, nn.BatchNorm2d(cout), nn.LeakyReLU())


class VAE(nn.Module):
    """Parameter container with the reference's module names (enc, fc_mu, fc_var, decoder_input, dec, final_layer)
    built in the reference's order (celeba_deepvanilla_vae.py:25-78) so that the initial values drawn from the
    global torch generator are identical."""

    def __init__(self, channels=1, h=None, w=None, zDim=2, dataset="mnist", latentspace_lr=1):
        super().__init__()
        self.channels, self.dataset, self.latentspace_lr, self.latentdim = channels, dataset, latentspace_lr, zDim
        c, blocks = channels, []
        for width in _WIDTHS:
            blocks.append(_stack("down", c, width))
            c = width
        self.enc = nn.Sequential(*blocks)
        self.fc_mu = nn.Linear(_WIDTHS[-1] * 4, zDim)
        self.fc_var = nn.Linear(_WIDTHS[-1] * 4, zDim)
        self.decoder_input = nn.Linear(zDim, _WIDTHS[-1] * 4)
        up = _WIDTHS[::-1]
        self.dec = nn.Sequential(*[_stack("up", up[i], up[i + 1]) for i in range(len(up) - 1)])
        last = _stack("up", up[-1], up[-1])
        self.final_layer = nn.Sequential(last[0], last[1], last[2],
                                         nn.Conv2d(up[-1], out_channels=self.channels, kernel_size=3, padding=1), nn.Sigmoid())


class ImgVAE(VAE):
    # behaviour switches set by the thin plugin modules (celeba_deepvanilla_vae, celeba_bettervanilla_vae, ...)
    PRIOR = "moment"            # 'moment' (deepvanilla :209) | 'sampled' (bettervanilla/celeba2cartoon/betaVAE)
    FORCE_BETA_ONE = True       # deepvanilla :185
    ALWAYS_FA_PRIOR = False     # celeba2cartoon: trainloop has no `favae` flag, the FA prior is always used
    UPDATE_X_USES_SELF_IMG = False   # celeba2cartoon/betaVAE iterate self.img whatever `img` is (:244)
    BATCH = 64                  # DataLoader(batch_size=64), :218
    # Grouped-batch mode (no counterpart in the reference; off = 1): GROUPS mini-batches of BATCH images per optimiser step, each
    # with BatchNorm statistics of its own, gradients averaged: the step a GROUPS-rank data-parallel run takes, on one GPU.  The
    # trajectory differs from the reference's (1 / GROUPS as many, larger steps); the visiting order and the noise are the same.
    GROUPS = int(os.environ.get("FAVAE_GROUPS", "1"))
    CHECKPOINT_INFIX = "_celeba_vanilla_"   # file name of restore_weights' checkpoint (:308-314); celeba2cartoon uses "8"
    # restore_weights keeps its best-ELBO checkpoint on the device; with write_checkpoints the file the reference writes
    # (./checkpoints/<dataset><infix><latentspace_lr>.pt, {epoch, elbo, state_dict, optimizer}) is also written, once per
    # trainloop (150 MB device -> host + serialisation, hence opt-in; FAVAE_WRITE_CHECKPOINTS=1 turns it on globally)
    write_checkpoints = os.environ.get("FAVAE_WRITE_CHECKPOINTS") == "1"

    def __init__(self, dimx=2, channels=1, var_x=0.01, lr=1e-5, h=None, w=None, dataset=None, latentspace_lr=1,
                 precision=None):
        # shape limits of the HIP executor (favae_vae_create): stated here so that they fail at construction, not at the
        # first GPU call
        if int(dimx) < 2 or int(dimx) % 2:
            raise ValueError("ImgVAE: the HIP executor needs an even latent size >= 2 (got dimx=%r)" % (dimx,))
        if not 1 <= int(channels) <= 4:
            raise ValueError("ImgVAE: the HIP executor handles 1..4 image channels (got %r)" % (channels,))
        if (h is not None and int(h) != 64) or (w is not None and int(w) != 64):
            raise ValueError("ImgVAE: the network is the reference's 64x64 architecture (got h=%r, w=%r)" % (h, w))
        super().__init__(channels=channels, h=h, w=w, zDim=dimx, dataset=dataset, latentspace_lr=latentspace_lr)
        self.lr, self.dimx, self.var_x = lr, dimx, var_x
        self.optim = torch.optim.Adam(self.parameters(), self.lr)
        self.device = _local_device()
        self.bce_loss = nn.BCELoss(reduction="sum")
        self.precision = precision or os.environ.get("FAVAE_CONV_DTYPE", "f32")
        self.img = None
        self.loss_during_training, self.elbo_training = [], []
        self.reconstruc_during_training, self.KL_QandP, self.KL_during_training = [], [], []
        self._eng = None
        self._synced_version = None
        # defer_pull: a coordinator that calls trainloop many times (SSHIBA.fit) sets it and calls sync() when it is done, so
        # the 150 MB of parameters + Adam moments are not copied back to the nn.Module after every trainloop
        self.defer_pull, self._stale = False, False
        self._img_key, self._data_dev = None, None
        self.x_mean_dev = self.x_cov_dev = None

    # ---- pickling: device state is rebuilt lazily --------------------------------------------------------------
    def __getstate__(self):
        self._pull()
        st = self.__dict__.copy()
        st["_stale"] = False
        for k in ("_eng", "_data_dev", "x_mean_dev", "x_cov_dev", "_ckp", "_img_src", "_comm_stream"):
            st[k] = None
        st["_synced_version"], st["_img_key"] = None, None
        return st

    # ---- module <-> executor synchronisation --------------------------------------------------------------------
    def _version(self):
        return sum(int(p._version) for p in self.parameters()) + sum(int(b._version) for b in self.buffers())

    def _engine(self):
        if self.device.type != "cuda":
            raise L.FavaeHipError("ImgVAE needs a GPU: the HIP path has no CPU fallback")
        if self._eng is None:
            from .engine import VaeEngine
            torch.cuda.set_device(self.device)
            self._eng = VaeEngine(self.dimx, self.channels, batch_max=self.BATCH * max(1, int(self.GROUPS)), dtype=self.precision,
                                  device=self.device)
            self._synced_version = None
        if self._synced_version != self._version():
            self._push()
        return self._eng

    def _push(self):
        e = self._eng
        sd = {k: v.detach().cpu() for k, v in super().state_dict().items()}
        e.load_state_dict(sd)
        names = dict(self.named_parameters())
        st = self.optim.state
        if len(st):
            m = {k: st[p]["exp_avg"].detach().cpu() if p in st else torch.zeros_like(p) for k, p in names.items()}
            v = {k: st[p]["exp_avg_sq"].detach().cpu() if p in st else torch.zeros_like(p) for k, p in names.items()}
            e.pack(m, e.M)
            e.pack(v, e.V)
            e.step = int(max(float(st[p]["step"]) for p in st))
        self._synced_version = self._version()
        self._stale = False

    def sync(self):
        """Bring the nn.Module parameters / buffers / torch optimiser state up to date with the executor (no-op if they are)."""
        if getattr(self, "_stale", False):
            self._pull()

    def state_dict(self, *a, **k):
        self.sync()
        return super().state_dict(*a, **k)

    def _pull(self):
        """Copy the trained parameters, BN buffers and Adam moments back into the nn.Module / torch optimiser."""
        e = self._eng
        if e is None:
            return
        sd = e.state_dict()
        with torch.no_grad():
            for k, p in self.named_parameters():
                p.copy_(sd[k])
            for k, b in self.named_buffers():
                if k in sd:
                    b.copy_(sd[k])
                elif k.endswith("num_batches_tracked"):
                    b.fill_(e.step)
        if e.step > 0:
            m, v = e.unpack(e.M), e.unpack(e.V)
            for k, p in self.named_parameters():
                self.optim.state[p] = {"step": torch.tensor(float(e.step)), "exp_avg": m[k].to(p.device),
                                       "exp_avg_sq": v[k].to(p.device)}
        self._synced_version = self._version()
        self._stale = False

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict)
        self._synced_version = None
        return out

    def to(self, *a, **k):       # parameters stay where the reference API put them; the executor owns device copies
        return self

    # ---- data residency ---------------------------------------------------------------------------------------
    def _source(self):
        """The training images as the caller handed them in: the array / tensor of the last trainloop / update_x call, or --
        on a model that came out of a pickle (cA/condition_a_pretrainedvae.py:87-94 un-pickles a trained ImgVAE and calls
        update_x() with no argument) -- `self.img`, the tensor the reference keeps (:187) and pickles with the model."""
        src = getattr(self, "_img_src", None)
        if src is None:
            src = self._img_src = self.img
        if src is None:
            raise ValueError("ImgVAE: no images yet (call trainloop(img=...) or update_x(img) first)")
        return src

    def _resident(self, img):
        """NHWC device copy of an image array / tensor; cached on the identity of the caller's object."""
        key = (id(img), tuple(img.shape))
        if key != self._img_key or self._data_dev is None:
            self._data_dev = self._engine().upload_images(img)
            self._img_key = key
        return self._data_dev

    # ---- reference API: network pieces ------------------------------------------------------------------------
    def _images_to_dev(self, x):
        x = x.detach().float().cpu().contiguous() if isinstance(x, torch.Tensor) else np.ascontiguousarray(x, dtype=np.float32)
        return self._engine().upload_images(x)

    def encoder(self, x):
        """(mu, std) of q(z|x) (:80-87); batch statistics when self.training, running statistics otherwise."""
        e = self._engine()
        d = self._images_to_dev(x)
        n = d.shape[0]
        mu = torch.empty(n, self.dimx, dtype=torch.float64, device=self.device)
        var = torch.empty_like(mu)
        for s in range(0, n, self.BATCH):
            b = min(self.BATCH, n - s)
            e.encode(d[s:s + b], None, b, mu[s:s + b], var[s:s + b], train=self.training)
        return mu.float(), var.sqrt().float()

    def reparameterize(self, mu, std):
        eps = torch.randn(std.shape)             # same stream as randn_like on the reference's CPU path (:91)
        return eps.to(std.device).mul(std).add_(mu)

    def decoder(self, z):
        """decoder(z) (:94-99): [n, zdim] -> [n, C, 64, 64] float32 on the device."""
        e = self._engine()
        zd = z.detach().float().to(self.device).contiguous()
        n = zd.shape[0]
        out = torch.empty(n, 64, 64, self.channels, dtype=torch.float32, device=self.device)
        for s in range(0, n, self.BATCH):
            b = min(self.BATCH, n - s)
            e.decode(zd[s:s + b], b, out[s:s + b], train=self.training)
        return out.permute(0, 3, 1, 2)

    def forward(self, x):
        mu, std = self.encoder(x)
        z = self.reparameterize(mu, std)
        return self.decoder(z), mu, std, z

    def kl_div(self, mu_Q, std_Q, mu_P, std_P):
        """KL(q||p) summed over z, mean over the batch (:137-150); convenience for callers, not on the hot path."""
        c = lambda t, lo, hi: torch.clamp(t, lo, hi)
        sq, sp = c(std_Q, 1e-6, 1e20), c(std_P, 1e-6, 1e20)
        vr = (sq / sp) ** 2
        t1 = ((c(mu_Q, -1e20, 1e20) - c(mu_P, -1e20, 1e20)) / sp) ** 2
        return (0.5 * (vr + t1 - 1 - vr.log())).sum(axis=1).mean()

    def gaussian_LL(self, true_images, reconstructed_images):
        D = int(np.prod(true_images.shape[1:]))
        x, r = true_images.reshape(-1, D), reconstructed_images.reshape(-1, D)
        return (-0.5 * (D * np.log(2 * np.pi) + D * np.log(self.var_x) + ((x - r) ** 2).sum(-1) / self.var_x)).mean()

    # ---- prior construction -----------------------------------------------------------------------------------
    def _prior_rows(self, Z, W, b):
        """Device fp64 [N, dimx] prior mean: Z W^T + b from moments (:209) or one scalar draw per factor (:150-154)."""
        from . import fa
        Zm, Wm, bm = Z["mean"], np.asarray(W["mean"]), np.asarray(b["mean"]).reshape(1, -1)
        shift = None
        if self.PRIOR == "sampled":
            rz, rw, rb = np.random.normal(), np.random.normal(), np.random.normal()
            shift = rz * np.sqrt(np.diag(np.asarray(Z["cov"])))
            Wm = Wm + rw * np.sqrt(np.diag(np.asarray(W["cov"])))
            bm = bm + rb * np.sqrt(np.diag(np.asarray(b["cov"])))
        Zd = Zm if isinstance(Zm, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(Zm)).to(self.device)
        Wt = np.ascontiguousarray(Wm.T)
        c = bm.ravel() if shift is None else bm.ravel() + shift @ Wt
        out = torch.empty(Zd.shape[0], self.dimx, dtype=torch.float64, device=self.device)
        fa.rows_gemm(Zd, torch.from_numpy(Wt).to(self.device), torch.from_numpy(np.ascontiguousarray(c)).to(self.device),
                     1.0, None, 0.0, out)
        return out

    # ---- trainloop (:173-295) ---------------------------------------------------------------------------------
    def trainloop(self, img=None, Z=None, W=None, b=None, tau=None, epochs=20, wandb=None, favae=False, beta=1):
        if self.FORCE_BETA_ONE:
            beta = 1
        if img is not None:
            self._img_src = img
            self.img = torch.from_numpy(img) if isinstance(img, np.ndarray) else torch.as_tensor(img)
        e = self._engine()
        data = self._resident(self._source())
        self.loss_during_training, self.elbo_training = [], []
        self.reconstruc_during_training, self.KL_during_training, self.KL_QandP = [], [], []
        if favae or self.ALWAYS_FA_PRIOR:
            prior = self._prior_rows(Z, W, b)
            tau_f = float(torch.Tensor(np.asarray(tau)).reshape(-1)[0])       # the fp32 cast of :210
        else:
            prior, tau_f = None, 1.0
        n_loc = data.shape[0]
        # Data-parallel epochs step in lockstep (one gradient all-reduce per optimiser step), so every rank visits the same
        # number of samples per epoch: the smallest shard's.  With np.array_split-style shards (dist.shard_bounds) that leaves
        # at most ONE sample per rank and epoch unvisited -- a different one each epoch, the permutation is drawn over the whole
        # shard -- and makes every mini-batch the same size on every rank, so the averaged gradient is exactly the gradient of the
        # global batch of BATCH * world_size samples.  (The FA statistics always run over all rows.)
        n = int(dist.min_scalar(n_loc, self.device))
        if dist.active() and dist.max_scalar(n_loc, self.device) - n >= self.BATCH and not getattr(self, "_warned_shards", False):
            self._warned_shards = True
            print("favae_hip: unbalanced shards: ranks with more than %d rows skip the surplus in each VAE epoch" % n)
        step_batch = self.BATCH * max(1, min(int(self.GROUPS), e.batch_max // self.BATCH))
        if dist.active() and dist.native_comm(self.device) is None:
            step_batch = self.BATCH     # (the torch.distributed rehearsal path below steps one mini-batch per rank)
        nb = e.epoch_steps(n, step_batch)
        metrics = torch.zeros(nb, 2, dtype=torch.float64, device=self.device)
        elbo_checker = -1e20

        def draw_epoch():
            """The random draws of one epoch, in the reference's order (:227-231 through the DataLoader and reparameterize :89-92)."""
            torch.empty((), dtype=torch.int64).random_()                      # DataLoader iterator base seed
            seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler seed
            g = torch.Generator()
            g.manual_seed(seed)
            perm = torch.randperm(n_loc, generator=g)[:n]
            eps = torch.cat([torch.randn(min(self.BATCH, n - s), self.dimx) for s in range(0, n, self.BATCH)], 0)
            return perm.to(torch.int32), eps

        # The draws of epoch e + 1 (65 ms of host time at N = 202 599: one randn per mini-batch from the global generator) are made
        # while the kernels of epoch e run: nothing else touches the generators between the epochs of one trainloop, so their order
        # is the reference's; the last epoch draws nothing ahead.
        drawn = draw_epoch() if int(epochs) > 0 else None
        for ep in range(int(epochs)):
            perm, eps = drawn
            perm_d, eps_d = perm.to(self.device), eps.to(self.device)
            if not dist.active():
                e.train_epoch(data, perm_d, eps_d, prior, tau_f, float(beta), float(self.var_x), float(self.lr), step_batch, metrics)
            elif dist.native_comm(self.device) is not None:
                # data parallel over the N shards, RCCL driven from the executor's own launch loop: one call per epoch
                # (bucketed all-reduce + bucket-wise Adam under the backward pass; favae_vae_train_epoch_dp)
                e.train_epoch_dp(dist.native_comm(self.device), data, perm_d, eps_d, prior, tau_f, float(beta), float(self.var_x),
                                 float(self.lr), step_batch, metrics)
                dist.mean_(metrics)
                dist.mean_(e.BUF)
            else:
                # the same through torch.distributed (gloo rehearsals, or when the native communicator is not available):
                # every rank runs its own mini-batch of 64, gradients are averaged before the (replicated) Adam step; BN
                # batch statistics stay per rank (group size 64, as :218)
                # The gradient arena is averaged in four buckets; the all-reduce of a bucket runs on a communication
                # stream while the layers behind it are still in their backward pass (decoder bucket under the whole
                # encoder backward).
                if getattr(self, "_comm_stream", None) is None:
                    self._comm_stream = torch.cuda.Stream(device=self.device)
                buckets = e.grad_buckets()
                for si, s in enumerate(range(0, n, self.BATCH)):
                    bsz = min(self.BATCH, n - s)
                    works, lo = [], 0
                    for a, b_, hi in buckets:
                        e.train_stages(data, perm_d[s:s + bsz], bsz, eps_d[s:s + bsz], prior, tau_f, float(beta),
                                       float(self.var_x), metrics[si], lo, hi)
                        works.append(dist.mean_async(e.G[a:b_], e, self._comm_stream))
                        lo = hi
                    for w in works:
                        if w is not None:
                            w.wait()
                    e.adam(float(self.lr))
                dist.mean_(metrics)
                dist.mean_(e.BUF)
            drawn = draw_epoch() if ep + 1 < int(epochs) else None              # (the epoch above is still running on the GPU)
            m = metrics.cpu().numpy()
            e.poll_error()
            rec, kl = m[:, 0], m[:, 1]
            loss = rec + beta * kl
            self.loss_during_training.append(np.float64(loss.sum() / nb))
            self.elbo_training.append(np.float64((-loss).sum() / nb))
            self.reconstruc_during_training.append(np.float64(rec.sum() / nb))
            self.KL_QandP.append(np.float64(kl.sum() / nb))
            elbo_checker = self.restore_weights(elbo_checker, ep)
            if wandb is not None:
                wandb.log({str(self.dataset) + " ELBO": float(self.elbo_training[-1]),
                           str(self.dataset) + " Gaussian LogLikelihood": float(-self.reconstruc_during_training[-1]),
                           str(self.dataset) + " KL(Q||P)": float(self.KL_QandP[-1])})
        self.training = True
        if self.defer_pull and not self.write_checkpoints:
            self._stale = True
        else:
            self._pull()
        if self.write_checkpoints and self.elbo_training and dist.rank() == 0:
            self.save_checkpoint()
        if not self.elbo_training:
            return True
        return abs(1 - np.mean(self.elbo_training[-10:]) / self.elbo_training[-1]) > 1e-4

    def restore_weights(self, elbo_checker, e):
        """Best-ELBO checkpoint / reload-on-regression (:297-329) as a device-side snapshot of parameters, BN running
        statistics and Adam state (what the reference round-trips through ./checkpoints/*.pt)."""
        eng = self._engine()
        if self.elbo_training[-1] > elbo_checker:
            self._ckp = dict(epoch=e + 1, elbo=self.elbo_training[-1], snap=eng.snapshot())
            elbo_checker = self.elbo_training[-1]
        if self.elbo_training[-1] < elbo_checker:
            eng.restore(self._ckp["snap"])
            elbo_checker = self._ckp["elbo"]
        return elbo_checker

    # ---- the reference's checkpoint file (:300-314, 319-328) ----------------------------------------------------
    def checkpoint_path(self):
        return "./checkpoints/" + str(self.dataset) + self.CHECKPOINT_INFIX + str(self.latentspace_lr) + ".pt"

    def save_checkpoint(self, path=None, epoch=None, elbo=None):
        """Write the current model + optimiser state in the layout of the reference's restore_weights file
        ({"epoch", "elbo", "state_dict", "optimizer"}); `elbo` is stored as a Python float so that the file loads with
        torch.load(weights_only=True).  After a trainloop the current state IS the best-ELBO state (every worse epoch was
        rolled back), i.e. what the reference's file holds."""
        path = self.checkpoint_path() if path is None else path
        self.sync()
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        ck = getattr(self, "_ckp", None) or {}
        elbo = float(elbo if elbo is not None else ck.get("elbo", self.elbo_training[-1] if self.elbo_training else float("nan")))
        epoch = int(epoch if epoch is not None else ck.get("epoch", len(self.elbo_training)))
        torch.save({"epoch": epoch, "elbo": elbo, "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()},
                    "optimizer": self.optim.state_dict()}, path)
        return path

    def load_checkpoint(self, path=None):
        """Load a restore_weights file (written by this build or by the reference's ImgVAE: same keys) into the module, the
        torch optimiser and -- at the next use -- the executor.  Tensors only: a file whose `elbo` is a pickled NumPy scalar
        (the reference stores np.float64) is refused by weights_only=True; re-save it with elbo=float(elbo) first."""
        path = self.checkpoint_path() if path is None else path
        ckp = torch.load(path, map_location="cpu", weights_only=True)
        self.load_state_dict(ckp["state_dict"])
        self.optim.load_state_dict(ckp["optimizer"])
        self._synced_version = None
        self._stale = False
        return ckp["epoch"], ckp["elbo"]

    # ---- update_x (:331-358) ----------------------------------------------------------------------------------
    def update_x(self, img=None):
        e = self._engine()
        torch.empty((), dtype=torch.int64).random_()        # the sequential DataLoader still draws its base seed
        if img is not None and getattr(self, "_img_src", None) is None and self.img is None:
            self._img_src = img
            self.img = torch.from_numpy(img) if isinstance(img, np.ndarray) else torch.as_tensor(img)
        if img is None or self.UPDATE_X_USES_SELF_IMG:
            data = self._resident(self._source())
        elif img is getattr(self, "_img_src", None):
            data = self._resident(img)
        else:
            data = self._images_to_dev(img)
        n = data.shape[0]
        self.x_mean_dev = torch.empty(n, self.dimx, dtype=torch.float64, device=self.device)
        self.x_cov_dev = torch.empty_like(self.x_mean_dev)
        e.encode_all(data, self.x_mean_dev, self.x_cov_dev, self.BATCH)
        self.training = False
        return self.x_mean_dev.cpu().numpy(), self.x_cov_dev.cpu().numpy()

    # ---- reconstruction (:360-385) ----------------------------------------------------------------------------
    def _latent_sample(self, mean, var):
        """torch.Tensor(mean + np.random.randn() * np.sqrt(var)) of :362 (one scalar draw; float64 arithmetic, float32 result);
        device tensors stay on the device."""
        if isinstance(mean, torch.Tensor):
            return (mean + np.random.randn() * var.sqrt()).float()
        return torch.Tensor(mean + np.random.randn() * np.sqrt(var))

    def reconstruction(self, mean, var):
        e = self._engine()
        sample = self._latent_sample(mean, var).to(self.device).contiguous()
        torch.empty((), dtype=torch.int64).random_()
        n = sample.shape[0]
        out = np.zeros((n, self.channels, 64, 64))
        buf = torch.empty(self.BATCH, 64, 64, self.channels, dtype=torch.float32, device=self.device)
        o64 = torch.empty(self.BATCH, self.channels, 64, 64, dtype=torch.float64, device=self.device)
        for s in range(0, n, self.BATCH):
            bsz = min(self.BATCH, n - s)
            e.decode(sample[s:s + bsz], bsz, buf)
            e.nhwc_to_nchw_f64(buf[:bsz], o64)
            out[s:s + bsz] = o64[:bsz].cpu().numpy()
        self.training = False
        return out

    def reconstruction_r2(self, mean, var):
        """r2_score(t.ravel(), reconstruction(mean, var).ravel()) of SSHIBA.update (cA/favae.py:839-840) computed on
        the device (same RNG consumption as reconstruction())."""
        e = self._engine()
        sample = self._latent_sample(mean, var).to(self.device).contiguous()
        torch.empty((), dtype=torch.int64).random_()
        acc = torch.zeros(3, dtype=torch.float64, device=self.device)
        e.reconstruct_stats(sample, self._resident(self._source()), acc, self.BATCH)
        dist.sum_(acc)
        sse, sy, syy = acc.cpu().numpy()
        cnt = dist.sum_scalar(float(sample.shape[0]), self.device) * 64 * 64 * self.channels
        self.training = False
        return 1.0 - sse / (syy - sy * sy / cnt)

    # ---- predict (:387-430) -----------------------------------------------------------------------------------
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
        torch.empty((), dtype=torch.int64).random_()
        was = self.training
        self.training = False
        img = self.decoder(rsample).double().cpu().numpy()
        self.training = was and False
        return img, rsample
