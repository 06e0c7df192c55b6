"""Process set-up of the training entry point: RNG seeding and device choice (the role of txt2vid/train/setup.py:7-31).

Seeding order matters for parity with the reference: Python's `random`, then numpy, then torch — the model constructors and
`init()` draw from torch's generator, `gen_perm` from numpy's, nothing on the hot path from `random`."""
import random

import numpy as np
import torch

from ..util.log import status, warn

_SEEDERS = (random.seed, np.random.seed, torch.manual_seed)


def set_seed(seed):
    """Seed all three generators with `seed` (drawn from `random` when None, like the reference) and return it."""
    seed = random.randint(1, 100000) if seed is None else seed
    for seeder in _SEEDERS:
        seeder(seed)
    return seed


def set_cuda(use_cuda=False):
    """The HIP device of this rank. There is no CPU training mode (the reference has none either, SURVEY §8a defect 8)."""
    if not use_cuda:
        if torch.cuda.is_available():
            warn('a GPU is visible but --cuda was not passed')
        raise SystemExit('the MI355X hot path has no CPU training mode: pass --cuda')
    from ..dist import local_device_index
    index = local_device_index()
    torch.cuda.set_device(index)
    return torch.device('cuda', index)


def setup(args):
    seed, device = set_seed(args.seed), set_cuda(use_cuda=args.cuda)
    status('Seed: %d' % seed)
    status('Device set to: %s' % device)
    return seed, device
