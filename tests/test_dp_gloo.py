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


def _bf16_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from txt2vid_amd import dist as tdist
    tdist.init_from_env('gloo')
    g = torch.Generator()
    g.manual_seed(5 + rank)
    w = torch.nn.Parameter(torch.zeros(4, 3, 3, 3))               # structurally sparse weight: taps 4 and 7 live
    b = torch.nn.Parameter(torch.zeros(37))                       # (an odd count: the compact part must still start aligned)
    gw = torch.zeros(4, 3, 9)
    gw[:, :, 4] = torch.randn(4, 3, generator=g)
    gw[:, :, 7] = torch.randn(4, 3, generator=g)
    w.grad = gw.view(4, 3, 3, 3).clone()
    b.grad = torch.randn(37, generator=g)
    arena = tdist.GradArena([w, b], live_taps={w: [4, 7]}, exchange_dtype=torch.bfloat16)

    class Opt(object):
        grad_scale = 1.0
    sync = tdist.make_grad_sync({'G': arena}, {'G': Opt()}, world)
    sync.time_exchanges(True)
    sync('G')
    ms = sync.exchange_ms()
    q.put((rank, w.grad.numpy().copy(), b.grad.numpy().copy(), gw.numpy().copy(), arena.exchanged_bytes(), ms))
    dist.destroy_process_group()


def test_bf16_gradient_exchange_world2():
    """The opt-in half-size exchange (SURVEY §8(e) "optionally reduce in bf16"; GradArena(exchange_dtype=torch.bfloat16) /
    T2V_GRAD_EXCHANGE=bf16): ONE collective over [dense | live taps] in bf16. The replicas receive bit-identical gradients, each
    within bf16 rounding of the fp32 sum (|err| <= 2^-8 (|a| + |b| + |a + b|)), dead taps stay exactly zero, the bytes halve —
    and the exchange is timed on host arenas too (GradSync.time_exchanges / exchange_ms)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, w0, b0, g0, n0, ms0), (r1, w1, b1, g1, n1, ms1) = res
    assert (w0 == w1).all() and (b0 == b1).all()                                 # replicas identical, bit for bit
    assert n0 == n1 == 2 * (37 + 4 * 3 * 2)
    want = torch.from_numpy(g0) + torch.from_numpy(g1)
    got = torch.from_numpy(w0).view(4, 3, 9)
    bound = 2.0 ** -8 * (torch.from_numpy(g0).abs() + torch.from_numpy(g1).abs() + want.abs()) + 1e-12
    assert ((got - want).abs() <= bound).all()
    dead = [t for t in range(9) if t not in (4, 7)]
    assert (got[:, :, dead] == 0).all()
    assert ms0['n'] == 1 and ms0['G'] > 0 and ms1['n'] == 1 and ms0['D'] == 0.0


def _shared_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from txt2vid_amd import dist as tdist
    tdist.init_from_env('gloo')
    # --end2end under data parallelism (train/gan.py): the text encoder's parameters sit in BOTH models' arenas
    dis, gen, enc = torch.nn.Linear(3, 2), torch.nn.Linear(3, 4), torch.nn.Linear(5, 3)
    for m in (dis, gen, enc):
        for p in m.parameters():
            torch.nn.init.constant_(p, 0.5)

    class Opt(object):
        grad_scale = 1.0
    arenas = {'D': tdist.model_arena([dis, enc]), 'G': tdist.model_arena([gen, enc])}
    sync = tdist.make_grad_sync(arenas, {'D': Opt(), 'G': Opt()}, world)
    x = torch.full((2, 5), float(rank + 1))
    out = {}
    for it in range(2):                                               # two iterations: the aliases of the first must not leak into the second
        # D step: backward through dis(enc(x)) -> exchange 'D' -> p.grad of the encoder aliases the D arena
        for m in (dis, enc):
            m.zero_grad(set_to_none=True)
        dis(enc(x)).sum().backward()
        sync('D')
        out['D%d' % it] = [p.grad.clone() for p in list(dis.parameters()) + list(enc.parameters())]
        d_alias = [p.grad.data_ptr() for p in enc.parameters()]
        # G step (cond_gan.py:90-118 zeroes G's and the encoder's gradients first): a fresh gradient for the encoder
        for m in (gen, enc):
            m.zero_grad(set_to_none=True)
        (gen(enc(x)) * 2.0).sum().backward()
        assert all(p.grad.data_ptr() != a for p, a in zip(enc.parameters(), d_alias))          # not written into the D arena
        keep = arenas['D'].flat.clone()
        sync('G')
        assert torch.equal(arenas['D'].flat, keep)                                               # the G exchange leaves D's arena alone
        out['G%d' % it] = [p.grad.clone() for p in list(gen.parameters()) + list(enc.parameters())]
    q.put((rank, {k: [t.numpy().copy() for t in v] for k, v in out.items()}))
    dist.destroy_process_group()


def test_end2end_shared_encoder_in_both_arenas_world2():
    """with --end2end the sentence encoder's parameters are members of the D arena AND the G arena. After the D
    exchange their p.grad aliases D's arena; the G step's zero_grad drops the alias, its backward produces fresh gradients and
    the G exchange sums THOSE — both exchanges give the all-rank sums, on both iterations, and neither arena is written through
    the other's views."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # reference: the same two backward passes per rank, summed by hand
    want = {}
    for rank in range(2):
        dis, gen, enc = torch.nn.Linear(3, 2), torch.nn.Linear(3, 4), torch.nn.Linear(5, 3)
        for m in (dis, gen, enc):
            for p in m.parameters():
                torch.nn.init.constant_(p, 0.5)
        x = torch.full((2, 5), float(rank + 1))
        dis(enc(x)).sum().backward()
        gD = [p.grad.clone() for p in list(dis.parameters()) + list(enc.parameters())]
        enc.zero_grad(set_to_none=True)
        (gen(enc(x)) * 2.0).sum().backward()
        gG = [p.grad.clone() for p in list(gen.parameters()) + list(enc.parameters())]
        for k, g in (('D', gD), ('G', gG)):
            want[k] = [a + b for a, b in zip(want[k], g)] if k in want else g
    for rank, out in res:
        for it in range(2):
            for k in ('D', 'G'):
                for got, w in zip(out['%s%d' % (k, it)], want[k]):
                    assert torch.allclose(torch.from_numpy(got), w), (rank, k, it)
