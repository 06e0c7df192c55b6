"""Which call sites still launch a single-weight pack kernel in a steady-state iteration (developer tool)."""
import collections
import os
import random
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd import _lib  # noqa: E402
from txt2vid_amd import functional as TF  # noqa: E402

dev = torch.device('cuda', 0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
prm = bench.Params()
pool = bench.synthetic_batches(8, 2, 100, dev)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
from txt2vid_amd.gan.trainer import train_iteration  # noqa: E402
for i in range(2):
    train_iteration(gan, pool[i % 2], None, optD, optG, losses, prm, dev)
L = _lib.lib()
sites = collections.Counter()
for name in ('t2v_pack_weight', 't2v_pack_weight_into'):
    orig = getattr(L, name)

    def wrap(*a, _o=orig, _n=name):
        st = traceback.extract_stack(limit=9)[:-1]
        sites[(_n, ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st[-6:])))] += 1
        return _o(*a)
    setattr(L, name, wrap)
train_iteration(gan, pool[0], None, optD, optG, losses, prm, dev)
torch.cuda.synchronize()
for k, v in sites.most_common(20):
    print(v, k)
