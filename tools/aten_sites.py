"""Which ATen kernels does one eager iteration of the benchmark workload launch, and under which autograd node?
(developer tool: `python tools/aten_sites.py`; needs the MI355X.) Prints, per ATen op that reached the GPU, its count and the
enclosing autograd-engine event (`...Backward`) or the Python frame inside txt2vid_amd that issued it."""
import collections
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd.gan.trainer import train_iteration  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
    gan = CondGan(gen=gen, discrims=[dis], cond_encoder=None, discrim_names=['video'], gp_scale=1.0)
    prm = bench.Params()
    prm.frame_sizes = [8, 16, 32, 64]
    pool = bench.synthetic_batches(32, 2, 100, dev, 64, 1)
    random.seed(100); np.random.seed(100); torch.manual_seed(100)
    for i in range(2):
        train_iteration(gan, pool[i % 2], None, optD, optG, losses, prm, dev)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_iteration(gan, pool[0], None, optD, optG, losses, prm, dev)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.name.startswith('aten::') and e.device_time_total > 0 and not e.cpu_children]
    sites = collections.Counter()
    for e in evs:
        p, chain = e.cpu_parent, []
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        node = next((n for n in chain if 'Backward' in n or 'AccumulateGrad' in n), None)
        frame = next((f for f in (e.stack or []) if 'txt2vid_amd' in f), None)
        sites[(e.name, node or '-', (frame or '-').split('/root/repo/')[-1][:90])] += 1
    for (name, node, frame), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print('%3d  %-22s %-60s %s' % (n, name, node[:60], frame))
    print('total ATen ops with GPU time:', len(evs))


if __name__ == '__main__':
    main()
