"""Which ATen kernels does one eager iteration of the benchmark workload launch, and under which autograd node?
(developer tool: `python tools/fan_in_sites.py`; needs the MI355X. `accum_sites.py` names the engine events of the adds that actually
launch; this one counts, on the recorded graphs, the node inputs that more than one edge reaches.) Prints, per ATen op that reached the GPU, its count and the
enclosing autograd-engine event (`...Backward`) or the Python frame inside txt2vid_amd that issued it."""
import collections
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from txt2vid_amd.gan.trainer import train_iteration  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)
    gan = CondGan(gen=gen, discrims=[dis], cond_encoder=None, discrim_names=['video'], gp_scale=1.0)
    prm = bench.Params()
    prm.frame_sizes = [8, 16, 32, 64]
    pool = bench.synthetic_batches(32, 2, 100, dev, 64, 1)
    random.seed(100); np.random.seed(100); torch.manual_seed(100)
    for i in range(2):
        train_iteration(gan, pool[i % 2], None, optD, optG, losses, prm, dev)
    torch.cuda.synchronize()
    # fan-in of every autograd node input: an input reached by k >= 2 edges costs k - 1 accumulation launches in the engine
    orig_backward = torch.Tensor.backward

    def fan_in(root, tag):
        seen, stack, indeg = set(), [root], collections.Counter()
        while stack:
            n = stack.pop()
            if n is None or id(n) in seen:
                continue
            seen.add(id(n))
            for nxt, idx in n.next_functions:
                if nxt is not None:
                    indeg[(id(nxt), idx, type(nxt).__name__)] += 1
                    stack.append(nxt)
        multi = collections.Counter()
        for (_, _, name), k in indeg.items():
            if k >= 2 and name != 'AccumulateGrad':
                multi[(name, k)] += 1
        print('--', tag, ': %d nodes; inputs with fan-in >= 2 (node type, fan-in) x count' % len(seen))
        for (name, k), c in sorted(multi.items(), key=lambda kv: -kv[1] * (kv[0][1] - 1)):
            print('   %3d x %-40s fan-in %d  -> %d adds' % (c, name, k, c * (k - 1)))

    def patched(self, *a, **kw):
        fan_in(self.grad_fn, 'backward of a %s scalar' % (tuple(self.shape),))
        return orig_backward(self, *a, **kw)
    torch.Tensor.backward = patched
    train_iteration(gan, pool[0], None, optD, optG, losses, prm, dev)
    torch.cuda.synchronize()
    torch.Tensor.backward = orig_backward
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        train_iteration(gan, pool[0], None, optD, optG, losses, prm, dev)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_stack_n=12):
        if not e.key.startswith('aten::') or getattr(e, 'device_time_total', 0) <= 0:
            continue
        frames = [f for f in (e.stack or []) if 'txt2vid_amd' in f or 'autograd' in f]
        rows.append((e.count, e.key, e.device_time_total, (frames[0] if frames else '-').split('/root/repo/')[-1][:110]))
    for n, name, us, frame in sorted(rows, key=lambda r: -r[0]):
        print('%3d  %-24s %8.1f us  %s' % (n, name, us, frame))
    print('total ATen ops with GPU time:', sum(r[0] for r in rows))


if __name__ == '__main__':
    main()
