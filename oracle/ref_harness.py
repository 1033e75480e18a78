"""Harness that imports the *real* reference (read-only, /root/reference) in this container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_goldens.py to produce the small fixtures under
tests/golden/.  Nothing here travels to the GPU box in executable form that matters: /root/reference does
not exist there, and no test, smoke() or bench.py leg imports this module at run time.

The shims are the ones SURVEY.md section 8(c) / Appendix A list; none of them touches a reference file:
  (i)   stub modules for the absent `wandb` and `pyro` packages (out-of-scope code paths only);
  (ii)  scipy.linalg.pinv(rcond=r) -> pinv(atol=r, rtol=0)  (scipy 1.7.3 semantics, environment.yml:118);
  (iii) torch.load(weights_only=False) for the reference's OWN freshly written checkpoint files;
  (iv)  a scratch cwd holding ./checkpoints ./logs ./plots ./results.
"""
import os
import sys
import types
import tempfile

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _WandbRun:
    def log(self, *a, **k):
        pass

    def finish(self, *a, **k):
        pass


def install_shims():
    import scipy.linalg
    import torch

    _stub("wandb", init=lambda *a, **k: _WandbRun())
    pyro = _stub("pyro")
    contrib = _stub("pyro.contrib")
    gp = _stub("pyro.contrib.gp")
    kernels = _stub("pyro.contrib.gp.kernels")
    dot = _stub("pyro.contrib.gp.kernels.dot_product", DotProduct=object)
    nnm = _stub("pyro.nn")
    mod = _stub("pyro.nn.module", PyroParam=object)
    pyro.contrib, contrib.gp, gp.kernels, kernels.dot_product = contrib, gp, kernels, dot
    pyro.nn, nnm.module = nnm, mod

    if not getattr(scipy.linalg.pinv, "_favae_shim", False):
        _orig = scipy.linalg.pinv

        def pinv(a, rcond=None, cond=None, **kw):
            r = rcond if rcond is not None else cond
            if r is not None:
                return _orig(a, atol=r, rtol=0.0, **kw)
            return _orig(a, **kw)

        pinv._favae_shim = True
        scipy.linalg.pinv = pinv

    if not getattr(torch.load, "_favae_shim", False):
        _tl = torch.load

        def load(*a, **k):
            k["weights_only"] = False  # only ever applied to files the reference wrote seconds earlier
            return _tl(*a, **k)

        load._favae_shim = True
        torch.load = load


def scratch_cwd():
    d = tempfile.mkdtemp(prefix="favae_ref_")
    for s in ("checkpoints", "logs", "plots", "results"):
        os.makedirs(os.path.join(d, s), exist_ok=True)
    os.chdir(d)
    return d


def import_reference(folder):
    """folder in {'condition_a_VAE', 'domain_adaptation', 'transfer_learning'}; returns the favae module."""
    install_shims()
    scratch_cwd()
    for k in ("favae", "celeba_deepvanilla_vae", "celeba_bettervanilla_vae", "celeba2cartoon_vae", "betaVAE"):
        sys.modules.pop(k, None)
    p = os.path.join(REF_ROOT, folder)
    sys.path[:] = [q for q in sys.path if not q.startswith(REF_ROOT)]
    sys.path.insert(0, p)
    import importlib

    favae = importlib.import_module("favae")
    if folder == "domain_adaptation":
        favae.img_vae.ImgVAE.KL_QandNormal = []
    return favae


def import_vae_module(folder, name):
    install_shims()
    scratch_cwd()
    sys.modules.pop(name, None)
    p = os.path.join(REF_ROOT, folder)
    sys.path[:] = [q for q in sys.path if not q.startswith(REF_ROOT)]
    sys.path.insert(0, p)
    import importlib

    return importlib.import_module(name)
