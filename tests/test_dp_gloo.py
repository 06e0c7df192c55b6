"""CPU, world_size 2 over gloo: the data-parallel gradient exchange of txt2vid_amd.dist, and the
global-batch semantics of SURVEY §8(e) (BCE terms are batch means, the multi-scale GP is a batch SUM and
is therefore scaled by world_size before gradient averaging) checked with the CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tganv2_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _arena_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from txt2vid_amd import dist as tdist
    r, w = tdist.init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2))]
    params[0].grad = torch.full((3, 4), float(rank + 1))
    params[1].grad = torch.arange(5.0) * (rank + 1)
    params[2].grad = None                                     # never reached on this rank: counts as zero

    class Opt(object):
        grad_scale = 1.0
    arena = tdist.GradArena(params)
    opt = Opt()
    sync = tdist.make_grad_sync({'D': arena}, {'D': opt}, world)
    sync('D')
    # (numpy through the queue: tensors travel as shared-memory handles, which a child that exits first may already have unlinked)
    q.put((rank, [p.grad.numpy().copy() for p in params], opt.grad_scale, arena.numel))
    dist.destroy_process_group()


def _sparse_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from txt2vid_amd import dist as tdist
    tdist.init_from_env('gloo')
    w = torch.nn.Parameter(torch.zeros(4, 3, 3, 3))               # conv weight whose gradient only lives in taps 4 and 7
    b = torch.nn.Parameter(torch.zeros(5))
    g = torch.zeros(4, 3, 9)
    g[:, :, 4] = rank + 1.0
    g[:, :, 7] = 10.0 * (rank + 1)
    w.grad = g.view(4, 3, 3, 3).clone()
    b.grad = torch.arange(5.0) + rank
    arena = tdist.GradArena([w, b], live_taps={w: [4, 7]})

    class Opt(object):
        grad_scale = 1.0
    sync = tdist.make_grad_sync({'G': arena}, {'G': Opt()}, world)
    sync('G')
    q.put((rank, w.grad.numpy().copy(), b.grad.numpy().copy(), arena.exchanged_bytes(), [p is w for p in arena.params]))
    dist.destroy_process_group()


def test_structurally_sparse_taps_exchange_world2():
    """Only the live taps of a structurally sparse weight gradient travel (ConvLSTM on a 1x1 map): same result as the dense
    exchange, a fraction of the bytes; dense parameters come first in the arena."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.zeros(4, 3, 9)
    want[:, :, 4], want[:, :, 7] = 3.0, 30.0
    for rank, gw, gb, nbytes, order in res:
        gw, gb = torch.from_numpy(gw), torch.from_numpy(gb)
        assert torch.equal(gw.view(4, 3, 9), want) and torch.equal(gb, torch.arange(5.0) * 2 + 1)
        assert nbytes == 4 * (5 + 4 * 3 * 2) and order == [False, True]


def test_grad_arena_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, grads, scale, numel in res:
        assert numel == 12 + 5 + 4 and scale == 0.5
        grads = [torch.from_numpy(g) for g in grads]
        assert torch.equal(grads[0], torch.full((3, 4), 3.0))            # 1 + 2, SUM; the optimiser applies 1/world
        assert torch.equal(grads[1], torch.arange(5.0) * 3)
        assert torch.equal(grads[2], torch.zeros(2, 2))


def test_dp_semantics_equal_global_batch_with_oracle():
    """Two replicas x local batch 2 with averaged gradients == one replica x global batch 4, provided the
    GP (a batch sum) is scaled by world_size locally. Single pyramid level, identical alphas."""
    torch.manual_seed(0)
    P = O.recipe_state(O.resnet3d_shapes('', 1, 64, 0))
    keys = [k for k, v in P.items() if v.dtype.is_floating_point]
    for k in keys:
        P[k].requires_grad_(True)
    real, fake = torch.randn(4, 1, 4, 8, 8), torch.randn(4, 1, 4, 8, 8)
    alpha = torch.rand(4, 1, 1, 1, 1)

    def loss_on(sl, gp_scale):
        r, f, a = real[sl], fake[sl], alpha[sl]
        ur, uf = O.resnet3d(P, r)[0], O.resnet3d(P, f)[0]
        return O.rsgan_discrim_loss(uf, ur) + 0.5 * gp_scale * O.gp_level(P, '', r, f, alpha=a)

    def grads(l):
        for k in keys:
            P[k].grad = None
        l.backward()
        return {k: (P[k].grad.clone() if P[k].grad is not None else torch.zeros_like(P[k])) for k in keys}
    g_global = grads(loss_on(slice(0, 4), 1.0))
    g0 = grads(loss_on(slice(0, 2), 2.0))
    g1 = grads(loss_on(slice(2, 4), 2.0))
    for k in keys:
        avg = 0.5 * (g0[k] + g1[k])
        assert torch.allclose(avg, g_global[k], rtol=2e-3, atol=1e-5 * max(1.0, float(g_global[k].abs().max()))), k


def _replica_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import random
    from txt2vid_amd import dist as tdist
    tdist.init_from_env('gloo')
    random.seed(1000 + 17 * rank)                        # no --seed: every rank would draw its own (train/setup.py:8)
    seed = tdist.broadcast_seed(random.randint(1, 100000))
    torch.manual_seed(12345 + rank)                      # ... and even with differently seeded constructors / init:
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 2))
    with torch.no_grad():
        net[1].running_mean.add_(float(rank))            # buffers travel too
    strided = torch.nn.Parameter(torch.randn(3, 3, 4, 5).permute(2, 3, 0, 1))        # a non-contiguous master copy
    holder = torch.nn.Module()
    holder.w = strided
    n = tdist.sync_replicas([net, None, holder])
    flat = torch.cat([t.detach().reshape(-1).double() for t in list(net.parameters()) + list(net.buffers()) + [strided]])
    q.put((rank, seed, n, flat.numpy().copy(), tdist.rank_world()))
    dist.destroy_process_group()


def test_replicas_identical_without_seed_world2():
    """no parameter broadcast existed and each rank drew its own seed. `broadcast_seed` + `sync_replicas`
    (train/gan.py) make the replicas bit-identical before the first step; rank 0 is the source."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, n0, f0, rw0), (r1, s1, n1, f1, rw1) = res
    assert s0 == s1 and n0 == n1 == 6 + 3 + 1        # conv / bn / linear weight + bias, 3 BatchNorm buffers, the strided tensor
    assert torch.equal(torch.from_numpy(f0), torch.from_numpy(f1))
    assert rw0 == (0, 2) and rw1 == (1, 2)
    random_state = __import__('random').Random(1000)
    assert s0 == random_state.randint(1, 100000)                   # rank 0's draw won
