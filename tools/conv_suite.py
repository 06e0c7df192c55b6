"""Kernel-time micro-benchmark over the convolution launch shapes of the benchmarked iteration (developer tool).
Times forward, data gradient and weight gradient of each shape with the library's own launch instrumentation
(hipExtLaunchKernelGGL start/stop events = kernel time only) and prints TFLOP/s on the executed (non-padding-tap) FLOPs.

    python tools/conv_suite.py [iters] [filter]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from txt2vid_amd import functional as TF          # noqa: E402
from txt2vid_amd._lib import lib                  # noqa: E402
import conv_cases as cc                           # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
flt = sys.argv[2] if len(sys.argv) > 2 else ''
dev = 'cuda:0'
torch.manual_seed(0)

D = cc.d_step_members
SUITE = [
    # name, Cin, Cout, kernel, members (N, D, H, W)
    ('D stem conv1 8m', 1, 64, (3, 3, 3), D(32, 0)),
    ('D stem conv2 8m', 64, 64, (3, 3, 3), D(32, 0)),
    ('D stem conv2 gp', 64, 64, (3, 3, 3), cc.gp_members(32, 0)),
    ('D stem conv2 Gstep', 64, 64, (3, 3, 3), cc.gp_members(32, 0) + cc.gp_members(32, 0)),
    ('D stem conv2 N64x4', 64, 64, (3, 3, 3), [(2 * n, d, h, w) for n, d, h, w in cc.gp_members(32, 0)]),
    ('D down0 conv1 8m', 64, 64, (3, 3, 3), D(32, 1)),
    ('X down0c1 1m 16^3', 64, 64, (3, 3, 3), [(12, 16, 16, 16)]),
    ('X down0c1 1m 2D64', 64, 64, (3, 3), [(12, 1, 64, 64)]),
    ('X down0c1 1m W4', 64, 64, (3, 3, 3), [(192, 16, 4, 4)]),
    ('X down0c1 lv3', 64, 64, (3, 3, 3), [(12, 4, 32, 32)]),
    ('D down0 conv2 8m', 64, 128, (3, 3, 3), D(32, 1)),
    ('D down1 conv1 8m', 128, 128, (3, 3, 3), D(32, 2)),
    ('D down1 conv2 8m', 128, 256, (3, 3, 3), D(32, 2)),
    ('D down2 conv1 8m', 256, 256, (3, 3, 3), D(32, 3)),
    ('D down2 conv2 8m', 256, 512, (3, 3, 3), D(32, 3)),
    ('D down3 conv2 8m', 512, 1024, (3, 3, 3), D(32, 4)),
    ('D down1 conv1 gp', 128, 128, (3, 3, 3), cc.gp_members(32, 2)),
    ('D down1 conv2 gp', 128, 256, (3, 3, 3), cc.gp_members(32, 2)),
    ('D down1 conv1 Gs', 128, 128, (3, 3, 3), cc.gp_members(32, 2) * 2),
    ('D down1 conv2 Gs', 128, 256, (3, 3, 3), cc.gp_members(32, 2) * 2),
    ('D down2 conv2 gp', 256, 512, (3, 3, 3), cc.gp_members(32, 3)),
    ('D down3 conv2 gp', 512, 1024, (3, 3, 3), cc.gp_members(32, 4)),
    ('G up2a 256->128 8x8', 256, 128, (3, 3), [(512, 1, 8, 8)]),
    ('G up1a 512->256 4x4', 512, 256, (3, 3), [(512, 1, 4, 4)]),
    ('G up0a 1024->512 2x2', 1024, 512, (3, 3), [(512, 1, 2, 2)]),
    ('G up2 128x128 8x8', 128, 128, (3, 3), [(512, 1, 8, 8)]),
    ('G up1 256x256 4x4', 256, 256, (3, 3), [(512, 1, 4, 4)]),
    ('G up0 512x512 2x2', 512, 512, (3, 3), [(512, 1, 2, 2)]),
    ('G blk1 64x64 16x16', 64, 64, (3, 3), [(128, 1, 16, 16)]),
    ('G blk2 32x32 32x32', 32, 32, (3, 3), [(32, 1, 32, 32)]),
]


def run(name, cin, cout, k, members):
    k3 = cc.k3(k)
    xs = [torch.randn(n, cin, d, h, w, device=dev) for n, d, h, w in members]
    gys = [torch.randn(n, cout, d, h, w, device=dev) for n, d, h, w in members]
    wt = torch.nn.Parameter(torch.randn(cout, cin, *k3, device=dev) * 0.05)
    out = (C.c_double * 18)()
    res = []
    for what, fn in (('fwd', lambda: TF.conv_group_raw(xs, wt, None, True, 0)),
                     ('dgrad', lambda: TF.conv_group_raw(gys, wt, None, False, 1, masks=xs)),
                     ('wgrad', lambda: TF.conv_group_wgrad_raw(xs, gys, tuple(wt.shape), True))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        lib().t2v_prof_begin(4096)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        lib().t2v_prof_end(out, 6)
        if what == 'wgrad':
            ms, fl, red = out[3] / iters, out[4] / iters, out[6] / iters
            res.append('%s %7.1f us %5.1f TF (+reduce %5.1f us)' % (what, ms * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0, red * 1e3))
        else:
            ms, fl, red = (out[0] + out[15]) / iters, (out[1] + out[16]) / iters, out[12] / iters      # fp32 GEMM or (bf16 mode) the bf16 GEMM
            res.append('%s %7.1f us %5.1f TF%s' % (what, ms * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0,
                                                   (' (+splitK %4.1f us)' % (red * 1e3)) if red > 0 else ''))
    M = sum(n * d * h * w for n, d, h, w in members)
    print('%-20s M=%6d %4d->%4d | %s' % (name, M, cin, cout, ' | '.join(res)), flush=True)


for case in SUITE:
    if flt in case[0]:
        run(*case)
