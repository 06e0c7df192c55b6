"""Phase ablations of conv_igemm_strip3_kernel<64> on the DownBlock-0 shapes (developer tool; DIAGNOSTIC build, wrong results).
  make -C txt2vid_amd/csrc ablation ; then on the GPU box, one process per flag set:
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=<bits> python tools/ablate_strip3.py
bits: 64 no global loads after the first round, 128 no LDS staging after the first round, 256 no barriers, 512 no MFMA loop,
1024 no epilogue stores."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from txt2vid_amd import functional as TF          # noqa: E402
import conv_cases as cc                           # noqa: E402

dev = 'cuda:0'
CASES = [('uniform 12x16^3 64->64 x27', 64, 64, (3, 3, 3), [(12, 16, 16, 16)]),
         ('ragged8 down0 conv1 64->64', 64, 64, (3, 3, 3), cc.d_step_members(32, 1)),
         ('G 128->128 8x8 x9', 128, 128, (3, 3), [(512, 1, 8, 8)]),
         ('G 256->256 4x4 x9', 256, 256, (3, 3), [(512, 1, 4, 4)])]
flags = os.environ.get('T2V_DEBUG_FLAGS', '0')
for name, cin, cout, k, members in CASES:
    xs = [torch.randn(n, cin, d, h, w, device=dev) for n, d, h, w in members]
    wt = torch.nn.Parameter(torch.randn(cout, cin, *cc.k3(k), device=dev) * 0.05)
    fl = sum(2.0 * n * d * h * w * cin * cout * (27 if (len(k) == 3 and d > 1) else 9) for n, d, h, w in members)
    for _ in range(5):
        TF.conv_group_raw(xs, wt, None, True, 0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            TF.conv_group_raw(xs, wt, None, True, 0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
    dt = min(ts)
    print('flags %5s  %-30s %7.1f us  %6.1f TF (nominal)' % (flags, name, dt * 1e6, fl / dt / 1e12), flush=True)
