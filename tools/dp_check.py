"""Two-rank rehearsal of the data-parallel path on ONE GPU (gloo instead of RCCL, both ranks on device 0):
the graph-replayed iteration with the gradient sink + the per-step arena all-reduce must leave every rank with the
same weights, and those must differ from what a rank would have learnt alone (the exchange is live).

    T2V_DIST_BACKEND=gloo T2V_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 \\
        --master-addr 127.0.0.1 --master-port 29531 tools/dp_check.py
"""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from txt2vid_amd import dist as tdist, functional as TF                         # noqa: E402
from txt2vid_amd.gan.trainer import GraphedTrainStep                            # noqa: E402

rank, world = tdist.init_from_env('nccl')
assert world == 2
dev = torch.device('cuda', tdist.local_device_index())
torch.cuda.set_device(dev)
B = 4


def run(sync):
    gen, dis, optD, optG, losses, CondGan = bench.build_models(dev)             # same seed -> identical replicas
    gan = CondGan(gen=gen, discrims=[dis], discrim_names=['video'], gp_scale=float(world if sync else 1))
    gs = None
    if sync:
        arenas = {'D': tdist.model_arena(dis, TF.copy_into), 'G': tdist.model_arena(gen, TF.copy_into)}
        gs = tdist.make_grad_sync(arenas, {'D': optD, 'G': optG}, world)
    pool = bench.synthetic_batches(B, 2, 100 + rank, dev)                       # different data per rank
    random.seed(100 + rank)
    np.random.seed(100 + rank)
    torch.manual_seed(100 + rank)
    step = GraphedTrainStep(gan, optD, optG, losses, bench.Params(), dev, tuple(pool[0].shape), grad_sync=gs, warmup=2)
    for i in range(5):                                                          # 2 eager, capture, 2 replays
        lD, lG = step.step(pool[i % 2])
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for p in list(dis.parameters()) + list(gen.parameters())])
    return chk, float(lD), float(lG)


chk, lD, lG = run(True)
both = [torch.empty_like(chk) for _ in range(world)]
dist.all_gather(both, chk.cpu().to(dev))
same = bool(torch.equal(both[0], both[1]))
solo, _, _ = run(False)
moved = float((solo - chk).abs().max())
if rank == 0:
    print('DP_CHECK replicas identical after 5 iterations: %s; differs from a solo run by %.3e; lossD %.5f lossG %.5f' %
          (same, moved, lD, lG))
    assert same and moved > 1e-6
dist.barrier()
dist.destroy_process_group()
