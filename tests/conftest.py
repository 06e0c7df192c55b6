import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # torch sizes its CPU pool by the machine's logical CPUs (128 threads on a GPU box whose cgroup grants 16 cores): the CPU oracle
    # these tests check against then runs 8x oversubscribed. Keep the pool inside the cores this process owns.
    try:
        import torch
        from txt2vid_amd.util.misc import host_threads
        if torch.get_num_threads() > host_threads():
            torch.set_num_threads(host_threads())
    except Exception:
        pass


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
