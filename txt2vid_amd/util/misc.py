import numpy as np


def gen_perm(n):
    """A random permutation of range(n) that is not the identity, drawn from numpy's global RNG —
    txt2vid/util/misc.py:3-8. n == 1 has no such permutation (the reference loops forever): raise."""
    if n < 2:
        raise ValueError('gen_perm needs n >= 2 (the reference never returns for n == 1)')
    old = np.array(range(n))
    new = np.random.permutation(old)
    while (new == old).all():
        new = np.random.permutation(old)
    return new


def count_params(model):
    return sum(p.numel() for p in model.parameters())
