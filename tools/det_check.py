"""Run-to-run determinism probe (developer tool): the same seeded iterations twice in one process."""
import sys, random, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_models_gpu as T
from txt2vid_amd.gan.trainer import train_iteration
DEV = 'cuda:0'


def batches():
    g = torch.Generator(); g.manual_seed(11)
    return [(torch.rand(4, 1, 16, 64, 64, generator=g) * 2 - 1).to(DEV) for _ in range(2)]


def seed():
    random.seed(5); np.random.seed(5); torch.manual_seed(5)


snaps = []
for run in range(3):
    gan, optD, optG, losses, prm = T._make_uncond()
    seed()
    snap = {}
    for i, x in enumerate(batches()):
        lD, lG, _, _ = train_iteration(gan, x, None, optD, optG, losses, prm, DEV)
        for tag, m in (('G', gan.gen), ('D', gan.discrims[0])):
            for k, p in m.named_parameters():
                snap['%d.%s.%s' % (i, tag, k)] = p.detach().double().sum().item()
                if p.grad is not None:
                    snap['%d.%s.%s.grad' % (i, tag, k)] = p.grad.detach().double().abs().sum().item()
        snap['%d.loss' % i] = (float(lD), float(lG))
    snaps.append(snap)
for run in (1,):
    diff = [k for k in snaps[0] if snaps[0][k] != snaps[run][k]]
    print('run0 vs run%d: %d differing entries of %d' % (run, len(diff), len(snaps[0])))
    d0 = [k for k in diff if k.startswith('0.D')]
    print('  step-0 D entries differing:', len(d0), d0[:10])
    g0 = [k for k in diff if k.startswith('0.G') and k.endswith('.grad')]
    same = [k for k in snaps[0] if k.startswith('0.G') and k.endswith('.grad') and k not in diff]
    print('  step-0 G grads differing:', len(g0), ' identical:', len(same), same[:20])
    for k in g0[-12:]:
        print('   ', k, snaps[0][k], snaps[run][k])
