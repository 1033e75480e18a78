"""oracle/ -- CPU restatement of the FA-VAE training hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import anything from here, and
only as the checker / the reported CPU baseline.  The product (fa-vae_amd/) never imports this package and
fails loudly when its HIP library is missing.

Contents
  fa_numpy.py      NumPy fp64 restatement of the SSHIBA mean-field updates (reference
                   condition_a_VAE/favae.py, domain_adaptation/favae.py; each function cites file:line).
  vae_torch.py     torch-CPU functional restatement of the conv VAE step / train loop / update_x
                   (reference condition_a_VAE/celeba_deepvanilla_vae.py and the sampled-prior variants).
  ref_harness.py   shims that let the *real* reference be imported in the build container.
  make_goldens.py  generates tests/golden/*.npz by running the real reference (build container only).

Pinning: the reference ships no tests, fixtures or golden vectors, so the oracle is
pinned against outputs of the reference itself run in the build container (make_goldens.py); the resulting
vectors are committed under tests/golden/ and re-checked by `pytest -m "not gpu"`.
"""
