"""Developer tool: every torch.empty / empty_like buffer handed to the kernels is pre-filled with NaN;
any kernel that reads a float it never wrote turns the losses / gradients into NaN."""
import sys, random, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
_e, _el = torch.empty, torch.empty_like


def pe(*a, **k):
    t = _e(*a, **k)
    return t.fill_(float('nan')) if t.is_floating_point() and t.is_cuda else t


def pel(*a, **k):
    t = _el(*a, **k)
    return t.fill_(float('nan')) if t.is_floating_point() and t.is_cuda else t


torch.empty, torch.empty_like = pe, pel
import test_models_gpu as T
from txt2vid_amd.gan.trainer import train_iteration
DEV = 'cuda:0'
gan, optD, optG, losses, prm = T._make_uncond()
random.seed(5); np.random.seed(5); torch.manual_seed(5)
g = torch.Generator(); g.manual_seed(11)
for i in range(2):
    x = (torch.rand(4, 1, 16, 64, 64, generator=g) * 2 - 1).to(DEV)
    lD, lG, _, _ = train_iteration(gan, x, None, optD, optG, losses, prm, DEV)
    print('step', i, float(lD), float(lG))
    for tag, m in (('G', gan.gen), ('D', gan.discrims[0])):
        bad = [k for k, p in m.named_parameters() if not torch.isfinite(p).all() or (p.grad is not None and not torch.isfinite(p.grad).all())]
        print('  ', tag, 'non-finite params/grads:', bad[:12], len(bad))
