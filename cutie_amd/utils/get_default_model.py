"""``get_default_model()`` -- hydra-free mirror of cutie/utils/get_default_model.py:14-28.

The reference composes eval_config with hydra (dataset d17-val => long-term memory off, mem_every=5), downloads
``cutie-base-mega.pth`` and returns ``CUTIE(cfg).cuda().eval()`` with the weights loaded.  There is no network here:
the checkpoint is read from ``$CUTIE_WEIGHTS`` or ``./weights/cutie-base-mega.pth`` when present, otherwise the model
keeps its seeded random initialisation (a warning is logged)."""
import logging
import os

import torch

from ..config import default_config
from ..model.cutie import CUTIE

log = logging.getLogger()


def get_default_model(weights: str = None, device: str = 'cuda') -> CUTIE:
    cfg = default_config()
    path = weights or os.environ.get('CUTIE_WEIGHTS') or os.path.join('weights', 'cutie-base-mega.pth')
    cfg['weights'] = path
    cutie = CUTIE(cfg).to(device).eval()
    if os.path.exists(path):
        cutie.load_weights(torch.load(path, map_location='cpu'))
    else:
        log.warning('%s not found: CUTIE keeps its random initialisation (no network access to download it)', path)
    return cutie
