"""Which Python call sites launch the plain element-wise kernels (t2v_add & co) during one eager iteration (developer tool)."""
import collections
import os
import random
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd import functional as TF  # noqa: E402
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
sites = collections.Counter()
orig = TF._ew


def spy(fn, name, *tensors):
    st = [f for f in traceback.extract_stack()[:-1] if 'txt2vid_amd' in f.filename][-4:]
    sites[(name, tuple(tensors[0].shape), ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(st)))] += 1
    return orig(fn, name, *tensors)


TF._ew = spy
ts.run(pool[0], None)
torch.cuda.synchronize()
for k, v in sites.most_common(30):
    print(v, k)
