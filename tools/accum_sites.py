"""Which backward nodes trigger the autograd engine's gradient accumulation (the ATen add / add_ launches that are left in the
replayed iteration): for every aten::add[_] of one steady-state eager iteration, the enclosing `evaluate_function: <Node>` event
and the tensor shape (developer tool)."""
import collections
import os
import random
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd.gan.trainer import TrainStep  # noqa: E402

dev = torch.device('cuda', 0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
prm = bench.Params()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pool = bench.synthetic_batches(B, 2, 100, dev)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
ts = TrainStep(gan, optD, optG, losses, prm, dev)
for i in range(2):
    ts.run(pool[i % 2], None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    ts.run(pool[0], None)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name not in ('aten::add', 'aten::add_'):
        continue
    p = ev.cpu_parent
    chain = []
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    node = next((c for c in chain if c.startswith('autograd::engine::evaluate_function')), chain[0] if chain else '?')
    sites[(node.replace('autograd::engine::evaluate_function: ', ''), ev.name, str(ev.input_shapes[0])[:40])] += 1
tot = sum(sites.values())
print('%d accumulation launches' % tot)
for k, v in sites.most_common(80):
    print(v, k)
