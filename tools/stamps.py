"""In-kernel timeline of the strip3 implicit GEMM (developer tool, DIAGNOSTIC build only).

Build (here, on the CPU box):  make -C txt2vid_amd/csrc stamps      -> tools/libt2v_stamps.so  (-DT2V_STAMPS)
Run (GPU box):                 T2V_LIB=tools/libt2v_stamps.so python tools/stamps.py [case ...]

Every wave of conv_igemm_strip3_kernel leaves one record (conv.hip, T2V_STAMPS): entry / loop start / loop end / exit ticks and the
summed ticks of the five phases of a barrier round. Read SHARES, not lengths (the stamps' fences forbid overlaps the product
kernel has). Prints, per case: launch span, spread of the workgroups' start times, workgroups per CU, and where a wave's life goes.
"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from txt2vid_amd import functional as TF          # noqa: E402
from txt2vid_amd import _lib                      # noqa: E402
import conv_cases as cc                           # noqa: E402

dev = 'cuda:0'
CASES = {
    'ragged8': (64, 64, (3, 3, 3), cc.d_step_members(32, 1)),
    'uniform': (64, 64, (3, 3, 3), [(12, 16, 16, 16)]),
    'gen128': (128, 128, (3, 3), [(512, 1, 8, 8)]),
    # the deep discriminator blocks: a few hundred voxels in all, K of several thousand (split-K launches)
    'deep_d3': (512, 1024, (3, 3, 3), cc.d_step_members(32, 4)),
    'deep_d2': (256, 512, (3, 3, 3), cc.d_step_members(32, 3)),
    'deep_d2c1': (256, 256, (3, 3, 3), cc.d_step_members(32, 3)),
}


def main():
    names = sys.argv[1:] or list(CASES)
    raw = C.CDLL(_lib.LIB_PATH)
    if not hasattr(raw, 't2v_stamps_set'):
        raise SystemExit('not a stamps build: set T2V_LIB=tools/libt2v_stamps.so')
    raw.t2v_stamps_set.argtypes = [C.c_void_p, C.c_uint]
    cap = 1 << 16
    buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
    assert raw.t2v_stamps_set(buf.data_ptr(), cap) == 0
    for name in names:
        cin, cout, k, members = CASES[name]
        xs = [torch.randn(n, cin, d, h, w, device=dev) for n, d, h, w in members]
        wt = torch.nn.Parameter(torch.randn(cout, cin, *cc.k3(k), device=dev) * 0.05)
        print('   plan:', cc.fwd_plan(members, cin, cout, k))
        for _ in range(3):
            TF.conv_group_raw(xs, wt, None, True, 0)
        torch.cuda.synchronize()
        buf.zero_()
        torch.cuda.synchronize()
        TF.conv_group_raw(xs, wt, None, True, 0)
        torch.cuda.synchronize()
        r = buf.cpu().numpy().reshape(cap, 16).astype(np.int64)
        r = r[r[:, 0] != 0]
        # (s_memtime counts per XCD: only differences inside one wave are meaningful; s_memrealtime (100 MHz) is chip-wide)
        life = r[:, 3] - r[:, 0]
        rt0 = r[:, 12].min()
        span_us = (r[:, 14].max() - rt0) / 100.0
        clk = np.median(life / np.maximum(r[:, 14] - r[:, 12], 1)) * 100.0          # ticks per us = MHz
        span = span_us * clk
        print('== %s: %d wave records, launch span %.1f us, in-kernel clock %.0f MHz (median ticks / realtime)' % (name, len(r), span_us, clk))
        st = (r[:, 12] - rt0) / 100.0
        print('   wave start after launch start: p50 %.2f  p90 %.2f  max %.2f us' % (np.percentile(st, 50), np.percentile(st, 90), st.max()))
        en = (r[:, 14] - rt0) / 100.0
        print('   wave exit: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us' % (np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max()))
        for g in sorted(set(r[:, 13].tolist())):
            m = r[:, 13] == g
            rr = r[m]
            rounds = rr[:, 9].mean()
            lf = (rr[:, 3] - rr[:, 0]).mean()
            print('   member %d: %5d waves, %4.1f rounds, life %7.0f (%.2f of span) | prologue %6.0f  loop %7.0f  epilogue %6.0f | per round: stage %5.0f  bar1 %5.0f  load %5.0f  mfma %6.0f  bar2 %5.0f' % (
                g, m.sum(), rounds, lf, lf / span, (rr[:, 1] - rr[:, 0]).mean(), (rr[:, 2] - rr[:, 1]).mean(), (rr[:, 3] - rr[:, 2]).mean(),
                (rr[:, 4] / rr[:, 9]).mean(), (rr[:, 5] / rr[:, 9]).mean(), (rr[:, 6] / rr[:, 9]).mean(), (rr[:, 7] / rr[:, 9]).mean(), (rr[:, 8] / rr[:, 9]).mean()))
        # residency: waves per (xcc, se, cu) -> workgroups per CU
        hw = r[:, 10]
        cu = (hw >> 8) & 0xf
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        xcc = r[:, 11] & 0xf
        key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        cnt = collections.Counter(key.tolist())
        hist = collections.Counter(v // 4 for v in cnt.values())
        print('   CUs used %d; workgroups per CU histogram: %s' % (len(cnt), dict(sorted(hist.items()))))
        simd = (hw >> 4) & 0x3
        print('   waves per SIMD id: %s' % dict(sorted(collections.Counter(simd.tolist()).items())))
        print('   mean wave life / span = %.2f' % (life.mean() / span))


if __name__ == '__main__':
    main()
