import os

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


def host_threads():
    """Cores this process may actually run on (cgroup quota / affinity aware), not the machine's logical CPU count."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                      # cgroup v2 quota, e.g. "1600000 100000" = 16 cores
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 64))


def limit_host_threads(reserve=4, cap=8):
    """Keep torch's CPU thread pool well inside the cores this process owns. On a box whose cgroup grants 16 of 256
    logical CPUs the default pool (one thread per logical CPU, spinning after every parallel region) starves the HIP
    runtime's submission thread: the replayed iteration took 41 ms instead of 21 ms (tools/cli_timing.py)."""
    import torch
    ranks_here = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))      # one process per GPU shares the node's cores
    n = max(1, min(cap, (host_threads() - reserve) // ranks_here))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return n
