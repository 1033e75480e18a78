"""Drop-in for the reference module of the same name (domain_adaptation/celeba2cartoon_vae.py).

Put this folder on sys.path in place of the reference folder; the trainers import it unchanged.  All compute runs in
libfavae_hip.so (see ../favae_hip/); this file only fixes the variant-specific behaviour.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from favae_hip.imgvae import VAE, ImgVAE as _ImgVAE  # noqa: E402,F401

torch.manual_seed(0)
np.random.seed(0)


class ImgVAE(_ImgVAE):
    PRIOR, FORCE_BETA_ONE, ALWAYS_FA_PRIOR, UPDATE_X_USES_SELF_IMG = "sampled", False, True, True   # :140-155, :244
    CHECKPOINT_INFIX = "8"                                                                         # :226

    def trainloop(self, img=None, Z=None, W=None, b=None, tau=None, epochs=20, beta=1, wandb=None):
        return super().trainloop(img=img, Z=Z, W=W, b=b, tau=tau, epochs=epochs, wandb=wandb, favae=True, beta=beta)
