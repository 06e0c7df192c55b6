"""Every C-ABI launch of one eager iteration by (entry point, Python call site): where do per-level / per-member launches remain?
(developer tool; `python tools/launch_sites.py [--cond]`)"""
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
from txt2vid_amd.gan.trainer import TrainStep  # noqa: E402

dev = torch.device('cuda', 0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
prm = bench.Params()
pool = bench.synthetic_batches(8, 2, 100, dev)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
ts = TrainStep(gan, optD, optG, losses, prm, dev)
for i in range(2):
    ts.run(pool[i % 2], None)
torch.cuda.synchronize()
cdll = _lib.lib()
sites = collections.Counter()
names = [n for n in _lib.SIGNATURES if not any(k in n for k in ('_floats', '_plan', '_ok', '_bytes', 'version', 'prof', '_splits'))]


def wrap(name, fn):
    def f(*a):
        st = [fr for fr in traceback.extract_stack()[:-1] if 'txt2vid_amd' in fr.filename][-3:]
        sites[(name, ' <- '.join('%s:%d' % (os.path.basename(fr.filename), fr.lineno) for fr in reversed(st)))] += 1
        return fn(*a)
    return f


for n in names:
    setattr(cdll, n, wrap(n, getattr(cdll, n)))
ts.run(pool[0], None)
torch.cuda.synchronize()
print(sum(sites.values()), 'C-ABI calls')
for k, v in sites.most_common(60):
    print(v, k)
