"""Phase ablations of conv_wgrad_thin_kernel on the full-clip stem (developer tool; DIAGNOSTIC build, wrong results).
  make -C txt2vid_amd/csrc ablation ; on the GPU box, one process per flag set:
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=<bits> python tools/ablate_thin.py
bits (after the first round of a workgroup): 64 no dL/dy loads, 128 no input gathers, 256 no LDS staging, 512 no MFMAs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from txt2vid_amd import functional as TF          # noqa: E402

dev = 'cuda:0'
flags = os.environ.get('T2V_DEBUG_FLAGS', '0')
for name, n, cin, cout, d, h, w in [('full-clip stem 1->64', 32, 1, 64, 16, 64, 64), ('full-clip skip 1x1 64ch', 32, 1, 64, 16, 64, 64)]:
    k = (3, 3, 3) if 'stem' in name else (1, 1, 1)
    x = torch.randn(n, cin, d, h, w, device=dev)
    gy = torch.randn(n, cout, d, h, w, device=dev)
    ws = (cout, cin) + k
    for _ in range(3):
        TF._wgrad_partial_launch([x], [gy], ws, False, True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            TF._wgrad_partial_launch([x], [gy], ws, False, True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    dt = min(ts)
    print('flags %5s  %-26s %7.1f us  %6.2f TB/s on dL/dy' % (flags, name, dt * 1e6, gy.numel() * 4 / dt / 1e12), flush=True)
