"""Timings of the rows either side of the hot path, next to the training iteration they feed:
  1. input pipeline: synthetic Moving-MNIST-shaped clips through DataLoader (pinned) + DevicePrefetcher, clips/s into HBM;
  2. sampling (`trainer.test` minus the PNG writer): eval-mode generator, full [B,1,16,64,64] clips per latent, videos/s,
     and the host oracle's eval forward on the box's cores for comparison;
  3. checkpoint: `CondGan.save_dict()` -> torch.save -> torch.load -> `load_from_dict`, seconds and MB.
  python tools/next_rows_bench.py [batch=32] [workers=8]"""
import io
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd import functional as TF  # noqa: E402
from txt2vid_amd.data import DevicePrefetcher, SyntheticMovingDigits, get_loader  # noqa: E402
from txt2vid_amd.util.misc import host_threads, limit_host_threads  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    limit_host_threads()
    dev = torch.device('cuda', 0)
    res = {'batch': B}

    # 1. input pipeline
    ds = SyntheticMovingDigits(length=B * 40, seed=3)
    loader = get_loader(ds, batch_size=B, num_workers=workers, has_captions=True)
    pre = DevicePrefetcher(loader, dev)
    x, _ = pre.next()
    n, t0 = 0, None
    while x is not None:
        x = TF.video_to_channel_first(x)
        if t0 is None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()                 # first batch = worker start-up
        else:
            n += x.shape[0]
        x, _ = pre.next()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res['input_pipeline'] = {'clips_per_s': n / dt, 'workers': workers, 'clip': '16x64x64x1 fp32 (262 KB)',
                             'what': 'SyntheticMovingDigits -> DataLoader(pin_memory) -> DevicePrefetcher -> channel-first in HBM'}

    # 2. sampling
    gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
    gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'])
    gen.eval()
    with torch.no_grad():
        z = torch.randn(B, gen.latent_size).to(dev)
        for _ in range(3):
            out = gan(z, cond=None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 20
        for _ in range(iters):
            out = gan(z, cond=None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    assert len(out) == 1 and tuple(out[0].shape) == (B, 1, 16, 64, 64)
    res['sampling'] = {'ms_per_batch': dt * 1e3, 'videos_per_s': B / dt, 'gflop_per_video_as_written': 9.95,
                       'tflops_as_written': 9.95 * B / dt / 1e3}
    from oracle import tganv2_oracle as O
    torch.set_num_threads(host_threads())
    PG = O.recipe_state(O.gen_shapes(num_channels=1))
    zc = torch.randn(4, 256)
    with torch.no_grad():
        O.multiscale_gen(PG, zc, None, training=False)
        t0 = time.perf_counter()
        O.multiscale_gen(PG, zc, None, training=False)
        dtc = time.perf_counter() - t0
    res['sampling']['cpu_oracle_videos_per_s'] = 4 / dtc
    res['sampling']['cpu_threads'] = host_threads()
    gen.train()

    # 3. checkpoint round trip
    t0 = time.perf_counter()
    buf = io.BytesIO()
    sd = gan.save_dict()
    sd.update({'optD': optD.state_dict(), 'optG': optG.state_dict()})
    torch.save(sd, buf)
    t_save = time.perf_counter() - t0
    buf.seek(0)
    t0 = time.perf_counter()
    back = torch.load(buf, map_location=dev, weights_only=False)
    gan.load_from_dict(back)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    res['checkpoint'] = {'mb': buf.getbuffer().nbytes / 1e6, 'save_s': t_save, 'load_s': t_load}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
