"""Seeding and device selection — txt2vid/train/setup.py:7-31 (same seed order: random, numpy, torch)."""
import os
import random

import numpy as np
import torch

from ..util.log import status, warn


def set_seed(seed):
    if seed is None:
        seed = random.randint(1, 100000)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


def set_cuda(use_cuda=False):
    if torch.cuda.is_available() and not use_cuda:
        warn('cuda is available')
    if not use_cuda:
        raise SystemExit('the MI355X hot path has no CPU training mode: pass --cuda (the reference has none either, SURVEY §8a defect 8)')
    from ..dist import local_device_index
    local = local_device_index()
    torch.cuda.set_device(local)
    return torch.device('cuda', local)


def setup(args):
    seed = set_seed(args.seed)
    device = set_cuda(use_cuda=args.cuda)
    status('Seed: %d' % seed)
    status('Device set to: %s' % device)
    return seed, device
