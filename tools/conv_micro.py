"""Micro-benchmark of the conv kernels on the dominant shapes (developer tool; also used for PMC runs).
  python tools/conv_micro.py [fwd|wgrad|both] [iters]"""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from txt2vid_amd import functional as TF

which = sys.argv[1] if len(sys.argv) > 1 else 'both'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = 'cuda:0'
torch.manual_seed(0)
# the 4 pyramid levels of the stem conv2 (64 -> 64, 3x3x3) at B=32 (x2: real || fake)
shapes = [(64, 64, 16, 8, 8), (32, 64, 8, 16, 16), (16, 64, 4, 32, 32), (8, 64, 2, 64, 64)]
xs = [torch.randn(s, device=dev) for s in shapes]
w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05)
gys = [torch.randn(s, device=dev) for s in shapes]
M = sum(s[0] * s[2] * s[3] * s[4] for s in shapes)
flops = 2.0 * M * 64 * 64 * 27


def timeit(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('%-28s %8.1f us  %6.1f TFLOP/s' % (name, dt * 1e6, flops / dt / 1e12))


if which in ('fwd', 'both'):
    timeit(lambda: TF.conv_group_raw(xs, w, None, False, 0), 'grouped fwd  M=%d' % M)
    timeit(lambda: TF.conv_group_raw(gys, w, None, False, 1), 'grouped dgrad M=%d' % M)
if which in ('wgrad', 'both'):
    timeit(lambda: TF.conv_group_wgrad_raw(xs, gys, (64, 64, 3, 3, 3)), 'grouped wgrad M=%d' % M)
