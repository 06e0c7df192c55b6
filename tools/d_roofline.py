"""CLI for txt2vid_amd/util/roofline.py (the north-star's D forward+backward roofline line).

    python tools/d_roofline.py [--batch 32] [--iters 5] [--no_attn]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--no_attn', action='store_true')
    a = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    from txt2vid_amd.util.roofline import d_fwdbwd_roofline
    print(json.dumps(d_fwdbwd_roofline(a.batch, a.iters, a.frames, a.size, not a.no_attn)))
