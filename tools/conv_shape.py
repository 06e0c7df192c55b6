"""Micro-benchmark of ONE conv shape (developer tool): forward, data gradient, weight gradient.
  python tools/conv_shape.py N Cin D H W Cout k [iters]        k = 1 | 3 (3 on every dim of extent > 1)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from txt2vid_amd import functional as TF

N, Cin, D, H, W, Cout, k = [int(v) for v in sys.argv[1:8]]
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 50
dev = 'cuda:0'
torch.manual_seed(0)
ks = (k if D > 1 or k == 1 else 1, k, k) if k == 3 else (1, 1, 1)
ks = (3 if (k == 3 and D > 1) else 1, k, k)
x = torch.randn(N, Cin, D, H, W, device=dev)
w = torch.nn.Parameter(torch.randn(Cout, Cin, *ks, device=dev) * 0.05)
gy = torch.randn(N, Cout, D, H, W, device=dev)
taps = sum(1 for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])
           if not ((D == 1 and a != ks[0] // 2) or (H == 1 and b != ks[1] // 2) or (W == 1 and c != ks[2] // 2)))
flops = 2.0 * N * D * H * W * Cin * Cout * taps


def timeit(fn, name):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('%-10s M=%-7d Cin=%-4d Cout=%-4d taps=%-2d %8.1f us  %6.1f TFLOP/s' % (name, N * D * H * W, Cin, Cout, taps, dt * 1e6, flops / dt / 1e12))


timeit(lambda: TF.conv_fwd_raw(x, w), 'fwd')
timeit(lambda: TF.conv_dgrad_raw(gy, w), 'dgrad')
timeit(lambda: TF.conv_wgrad_raw(x, gy, tuple(w.shape)), 'wgrad')
