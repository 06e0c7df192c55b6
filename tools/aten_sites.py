"""Python call sites of the ATen (non-t2v) device ops inside one steady-state eager iteration (developer tool)."""
import collections
import os
import random
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd import dist as tdist  # noqa: E402
from txt2vid_amd import functional as TF  # noqa: E402

COND = '--cond' in sys.argv          # the text-conditioned iteration (BASELINE configs[2]; `--bf16` as there)
if '--bf16' in sys.argv:
    TF.set_conv_precision('bf16')
dev = torch.device('cuda', 0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev, cond=COND)
txt = None
codes = None
if COND:
    from txt2vid_amd.data import Vocab  # noqa: E402
    from txt2vid_amd.models.txt.basic import Seq2Seq  # noqa: E402
    from txt2vid_amd.util.torch.init import init  # noqa: E402
    txt = Seq2Seq(vocab_size=len(Vocab()))
    init(txt, 'xavier')
    txt.to(dev)
    tokens = torch.randint(4, len(Vocab()), (8, 8)).to(dev)
    codes = txt.encode(tokens, [8] * 8)[2].detach()
gan = CondGan(gen=gen, discrims=[dis], cond_encoder=txt, discrim_names=['video'])
prm = bench.Params()
pool = bench.synthetic_batches(8, 2, 100, dev)
random.seed(1); np.random.seed(1); torch.manual_seed(1)
from txt2vid_amd.gan.trainer import TrainStep  # noqa: E402
ts = TrainStep(gan, optD, optG, losses, prm, dev)
for i in range(2):
    ts.run(pool[i % 2], codes)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    ts.run(pool[0], codes)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    n = ev.name
    if n in ('aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::zeros_like', 'aten::clone',
             'aten::mul', 'aten::sum', 'aten::contiguous', 'aten::ones_like', 'aten::empty_like') and n not in ('aten::empty_like',):
        shapes = str(ev.input_shapes)[:60]
        st = [s for s in (ev.stack or []) if 'txt2vid_amd' in s or 'autograd' in s][:3]
        sites[(n, shapes, ' <- '.join(os.path.basename(s.split(',')[0]) + ':' + s.split('(')[-1].split(')')[0] if '(' in s else s for s in st))] += 1
for k, v in sites.most_common(45):
    print(v, k)
