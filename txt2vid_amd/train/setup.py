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


def get_rng_state():
    """The three host generators every per-iteration draw comes from (z, sub-sample phases, GP alphas: torch; caption
    permutations: numpy; nothing on the hot path: `random`) as one picklable dict — saved next to the weights so that a resumed
    run continues the draw sequence (the reference's checkpoints carry no generator state, trainer.py:269-279)."""
    ns = np.random.get_state()
    return {'python': random.getstate(), 'numpy': (ns[0], ns[1].tolist(), int(ns[2]), int(ns[3]), float(ns[4])),
            'torch': torch.get_rng_state().clone()}


def set_rng_state(state):
    """Inverse of `get_rng_state` (a dict missing a generator leaves that generator alone)."""
    if not state:
        return
    if 'python' in state:
        py = state['python']
        random.setstate((py[0], tuple(py[1]), py[2]))
    if 'numpy' in state:
        n = state['numpy']
        np.random.set_state((n[0], np.asarray(n[1], dtype=np.uint32), n[2], n[3], n[4]))
    if 'torch' in state:
        torch.set_rng_state(torch.as_tensor(state['torch'], dtype=torch.uint8).cpu())
