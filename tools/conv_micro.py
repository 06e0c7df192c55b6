"""Micro-benchmark of the conv kernels on the dominant shapes (developer tool; also the workload of the PMC passes behind
profiles/r03_pmc_traffic.json / r03_pmc_sq.json).
  python tools/conv_micro.py [fwd|wgrad|pool|both] [iters]
fwd / wgrad: the DownBlock-0 first convolution (64 -> 64, 3x3x3) over the 8 discriminator-step members at B=32 (M = 49 152 voxels) —
the launch shape of the iteration's dominant kernel, conv_igemm_strip3_kernel<64,32,2,true>, and conv_wgrad3_kernel;
pool: the stem's pooled second convolution (64 -> 64) over its 8 members (393 216 voxels -> 49 152 pooled rows): box-sum, pooled
forward / data gradient (+ unbox) / weight gradient."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from txt2vid_amd import functional as TF          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'both'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = 'cuda:0'
torch.manual_seed(0)
B = 32


def stage(s):
    out = []
    for lvl in range(4):
        b, t, sz = -(-B // (1 << lvl)), 16 >> lvl, 8 << lvl
        for _ in range(s):
            t, sz = (t + 1) // 2 if t > 1 else 1, max(1, sz // 2)
        out.append((b, t, sz, sz))
    return [(2 * n, d, h, w) for n, d, h, w in out] + out


def timeit(fn, name, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('%-44s %8.1f us  %6.1f TFLOP/s' % (name, dt * 1e6, flops / dt / 1e12))


w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05)
if which in ('fwd', 'wgrad', 'both'):
    mem = stage(1)                                                    # down0 conv1: [.,64,8,4,4] ... [.,64,1,32,32]
    xs = [torch.randn(n, 64, d, h, w_, device=dev) for n, d, h, w_ in mem]
    gys = [torch.randn_like(x) for x in xs]
    M = sum(x.numel() // 64 for x in xs)
    flops = sum(2.0 * (x.numel() // 64) * 64 * 64 * (27 if x.shape[2] > 1 else 9) for x in xs)
    if which in ('fwd', 'both'):
        timeit(lambda: TF.conv_group_raw(xs, w, None, True, 0), 'down0 conv1 fwd   M=%d (strip3<64>)' % M, flops)
        timeit(lambda: TF.conv_group_raw(gys, w, None, False, 1, masks=xs), 'down0 conv1 dgrad M=%d (strip3<64>)' % M, flops)
    if which in ('wgrad', 'both'):
        timeit(lambda: TF.conv_group_wgrad_raw(xs, gys, (64, 64, 3, 3, 3), True), 'down0 conv1 wgrad M=%d (wgrad3 + reduce)' % M, flops)
if which in ('pool', 'both'):
    mem = stage(0)                                                    # stem conv2 members: [.,64,16,8,8] ... [.,64,2,64,64]
    hs = [torch.randn(n, 64, d, h, w_, device=dev) for n, d, h, w_ in mem]
    tm = [TF.pool_tmode(h.shape, True) for h in hs]
    shapes = [tuple(h.shape) for h in hs]
    rts = TF.boxsum_raw(hs, tm, True)
    zs = TF.pool_fwd_raw(rts, shapes, tm, w, None)
    gz = [torch.randn_like(z) for z in zs]
    Mp = sum(z.numel() // 64 for z in zs)
    flops = 2.0 * Mp * 64 * 64 * 27
    timeit(lambda: TF.boxsum_raw(hs, tm, True), 'stem box-sum (393216 voxels)', 0.0)
    timeit(lambda: TF.pool_fwd_raw(rts, shapes, tm, w, None), 'stem pooled fwd   M\'=%d (conv_pool_fwd)' % Mp, flops)
    planes = TF.pool_dgrad_raw(gz, shapes, tm, w)
    timeit(lambda: TF.pool_dgrad_raw(gz, shapes, tm, w), 'stem pooled dgrad (conv_pool_dgrad)', flops)
    timeit(lambda: TF.unbox_raw(planes, shapes, tm, masks=hs), 'stem unbox (+ReLU mask)', 0.0)
    timeit(lambda: TF.pool_wgrad_raw(rts, gz, shapes, tm, (64, 64, 3, 3, 3)), 'stem pooled wgrad (conv_pool_wgrad + reduce)', flops)
