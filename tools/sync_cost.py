"""Developer check: wall time per graph-replayed iteration with and without a host sync after every iteration."""
import sys
import time

import torch

sys.path.insert(0, '.')
from txt2vid_amd.gan.trainer import GraphedTrainStep
import bench

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
pool = bench.synthetic_batches(32, 2, 1, dev)
g = GraphedTrainStep(gan, optD, optG, losses, bench.Params(), dev, tuple(pool[0].shape), warmup=2)
for i in range(5):
    g.step(pool[i % 2])
torch.cuda.synchronize()
for mode in ('no sync', 'sync every iteration', 'read loss one iteration late', 'sync + 2 ms idle', 'sync + 8 ms idle', 'sync + 8 ms busy host'):
    t0 = time.perf_counter()
    th = 0.0
    prev = None
    for i in range(20):
        a = time.perf_counter()
        lD, lG = g.step(pool[i % 2])
        th += time.perf_counter() - a
        if mode.startswith('sync'):
            float(lD)
            if 'idle' in mode:
                time.sleep(0.002 if '2 ms' in mode else 0.008)
            elif 'busy' in mode:
                e = time.perf_counter() + 0.008
                while time.perf_counter() < e:
                    pass
        elif mode.startswith('read'):
            cur = lD.clone()
            if prev is not None:
                float(prev)
            prev = cur
    torch.cuda.synchronize()
    print('%-32s %.2f ms / iteration (host time inside step(): %.2f ms)' % (mode, (time.perf_counter() - t0) / 20 * 1e3, th / 20 * 1e3))
