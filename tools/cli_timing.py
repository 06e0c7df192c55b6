"""Developer check: per-iteration wall time of the CLI training loop pieces (loader, channel-first copy, step)."""
import sys
import time

import torch

sys.path.insert(0, '.')
from txt2vid_amd import data, functional as TF
from txt2vid_amd.gan.trainer import GraphedTrainStep
import bench

dev = torch.device('cuda', 0)
if len(sys.argv) > 3:
    torch.set_num_threads(int(sys.argv[3]))
torch.cuda.set_device(0)
gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
prm = bench.Params()
ds = data.my_dataset(data='synthetic', num_frames=16, length=4096, size=64, channels=1, seed=1)
loader = data.get_loader(dset=ds, batch_size=32, num_workers=int(sys.argv[1]) if len(sys.argv) > 1 else 4)
if len(sys.argv) > 2 and sys.argv[2] == 'nopin':
    loader = torch.utils.data.DataLoader(ds, batch_size=32, shuffle=True, num_workers=int(sys.argv[1]), collate_fn=data.collate_fn, drop_last=True, pin_memory=False)
pre = data.DevicePrefetcher(loader, dev)
g = None
t_load = t_cf = t_step = t_gpu = t_host = t_poll = 0.0
n = 0
x, y = pre.next()
while x is not None and n < 60:
    t0 = time.perf_counter()
    x = TF.video_to_channel_first(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if g is None:
        g = GraphedTrainStep(gan, optD, optG, losses, prm, dev, tuple(x.shape), warmup=2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lD, lG = g.step(x)
    e1.record()
    th = time.perf_counter()
    while not e1.query():
        pass
    tq = time.perf_counter()
    float(lD)
    t2 = time.perf_counter()
    if n >= 10:
        t_poll += tq - th
    if n >= 10:
        t_gpu += e0.elapsed_time(e1)
        t_host += th - t1
    x, y = pre.next()
    t3 = time.perf_counter()
    if n >= 10:
        t_cf += t1 - t0
        t_step += t2 - t1
        t_load += t3 - t2
    n += 1
k = n - 10
print('per iteration: channel-first %.2f ms, step %.2f ms (host part %.2f ms, polling e1 %.2f ms, GPU events %.2f ms), next batch wait %.2f ms (graphs captured: %s)' %
      (t_cf / k * 1e3, t_step / k * 1e3, t_host / k * 1e3, t_poll / k * 1e3, t_gpu / k, t_load / k * 1e3, g.graphs is not None))
