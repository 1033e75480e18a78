"""Drop-in for the reference module of the same name (transfer_learning/betaVAE.py).

Put this folder on sys.path in place of the reference folder; the trainers import it unchanged.  All compute runs in
libfavae_hip.so (see ../favae_hip/); this file only fixes the variant-specific behaviour.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from favae_hip.imgvae import VAE, ImgVAE as _ImgVAE  # noqa: E402,F401


class ImgVAE(_ImgVAE):
    PRIOR, FORCE_BETA_ONE, UPDATE_X_USES_SELF_IMG = "sampled", False, True        # :144-156, :245

    def trainloop(self, img=None, Z=None, W=None, b=None, tau=None, epochs=20, wandb=None, beta=1, favae=False):
        return super().trainloop(img=img, Z=Z, W=W, b=b, tau=tau, epochs=epochs, wandb=wandb, favae=favae, beta=beta)
