"""Drop-in for the reference module of the same name (domain_adaptation/favae.py).

Put this folder on sys.path in place of the reference folder; the trainers import it unchanged.  All compute runs in
libfavae_hip.so (see ../favae_hip/); this file only fixes the variant-specific behaviour.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import celeba2cartoon_vae as img_vae  # noqa: E402   (reference :20 -- change only this line to change the VAE plugin)
from favae_hip.sshiba import SSHIBA as _SSHIBA, HyperParameters, Qdistribution  # noqa: E402,F401

torch.manual_seed(0)      # reference :23-24
np.random.seed(0)


class SSHIBA(_SSHIBA):
    VARIANT = "B"
    img_vae_module = img_vae
