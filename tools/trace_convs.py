"""Developer tool: log every grouped conv / wgrad call of ONE training iteration (phase, mode, members, voxels)."""
import sys, random, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_models_gpu as T
from txt2vid_amd import functional as TF
from txt2vid_amd.gan.trainer import TrainStep
log = []
_raw, _wg = TF.conv_group_raw, TF.conv_group_wgrad_raw
phase = ['?']
def raw(xs5, w5, bias=None, relu_in=False, mode=0):
    log.append((phase[0], 'dgrad' if mode else 'fwd', len(xs5), sum(t.shape[0] * t.shape[2] * t.shape[3] * t.shape[4] for t in xs5), tuple(w5.shape)))
    return _raw(xs5, w5, bias, relu_in, mode)
def wg(xs5, gys5, wshape, relu_in=False):
    log.append((phase[0], 'wgrad', len(xs5), sum(t.shape[0] * t.shape[2] * t.shape[3] * t.shape[4] for t in xs5), tuple(wshape)))
    return _wg(xs5, gys5, wshape, relu_in)
TF.conv_group_raw, TF.conv_group_wgrad_raw = raw, wg
gan, optD, optG, losses, prm = T._make_uncond()
B = 32
x = (torch.rand(B, 1, 16, 64, 64) * 2 - 1).to('cuda:0')
ts = TrainStep(gan, optD, optG, losses, prm, 'cuda:0')
phase[0] = 'D'; ts.part_d(x, None)
phase[0] = 'G'; ts.part_g(); ts.part_end()
import collections
c = collections.Counter((p, k, n, m, w) for p, k, n, m, w in log if w[:2] == (64, 64) and len(w) == 5 and w[2] == 3)
for k, v in sorted(c.items()):
    print(v, k)
print('---- backward detail of the stem conv2 node')
log.clear()
orig = TF.ConvG.backward
def bw(ctx, *gys):
    w = ctx.saved_tensors[0]
    if tuple(w.shape) == (64, 64, 3, 3, 3) and ctx.saved_tensors[1].shape[-1] >= 8 and len(gys) == 8:
        print('ConvG.backward live', [g is not None for g in gys], 'needs', ctx.needs_input_grad[3:], 'grad_enabled', torch.is_grad_enabled())
    return orig(ctx, *gys)
TF.ConvG.backward = staticmethod(bw)
gan, optD, optG, losses, prm = T._make_uncond()
ts = TrainStep(gan, optD, optG, losses, prm, 'cuda:0')
phase[0] = 'D'; ts.part_d(x, None)
print('-- G')
ts.part_g()
