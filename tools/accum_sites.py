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

COND = '--cond' in sys.argv           # the text-conditioned iteration (BASELINE configs[2])
args_ = [a for a in sys.argv[1:] if not a.startswith('--')]
dev = torch.device('cuda', 0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev, cond=COND)
B = int(args_[0]) if args_ else 8
txt = codes = None
if COND:
    from txt2vid_amd.data import Vocab  # noqa: E402
    from txt2vid_amd.models.txt.basic import Seq2Seq  # noqa: E402
    from txt2vid_amd.util.torch.init import init  # noqa: E402
    txt = Seq2Seq(vocab_size=len(Vocab()))
    init(txt, 'xavier')
    txt.to(dev)
    codes = txt.encode(torch.randint(4, len(Vocab()), (B, 8)).to(dev), [8] * B)[2].detach()
gan = CondGan(gen=gen, discrims=[dis], cond_encoder=txt, discrim_names=['video'])
prm = bench.Params()
pool = bench.synthetic_batches(B, 2, 100, dev)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
ts = TrainStep(gan, optD, optG, losses, prm, dev)
for i in range(2):
    ts.run(pool[i % 2], codes)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    ts.run(pool[0], codes)
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
