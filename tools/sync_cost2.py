import sys, time
import torch
sys.path.insert(0, '.')
from txt2vid_amd import data, functional as TF
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
host = [p.permute(0, 2, 1, 3, 4).contiguous().cpu() for p in pool]        # [B,T,C,H,W] host batches


def run(name, get):
    t0 = time.perf_counter()
    tf = 0.0
    for i in range(20):
        x = get(i)
        lD, lG = g.step(x)
        a = time.perf_counter()
        float(lD)
        tf += time.perf_counter() - a
    torch.cuda.synchronize()
    print('%-44s %.2f ms / iteration (float(lD): %.2f ms)' % (name, (time.perf_counter() - t0) / 20 * 1e3, tf / 20 * 1e3))


run('device-resident batches', lambda i: pool[i % 2])
run('pageable host batch .to(device)', lambda i: TF.video_to_channel_first(host[i % 2].to(dev)))
pinned = [h.pin_memory() for h in host]
run('pinned host batch .to(device, non_blocking)', lambda i: TF.video_to_channel_first(pinned[i % 2].to(dev, non_blocking=True)))
side = torch.cuda.Stream()


def via_side(i):
    with torch.cuda.stream(side):
        x = pinned[i % 2].to(dev, non_blocking=True)
    torch.cuda.current_stream().wait_stream(side)
    x.record_stream(torch.cuda.current_stream())
    return TF.video_to_channel_first(x)


run('pinned, copied on a side stream', via_side)
ds = data.my_dataset(data='synthetic', num_frames=16, length=4096, size=64, channels=1, seed=1)
t0 = time.perf_counter()
b = data.collate_fn([ds[j] for j in range(32)])
print('generating + collating one synthetic batch on the host: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
