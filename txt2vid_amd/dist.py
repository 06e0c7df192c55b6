"""Data parallelism for the hot path: one process per GPU, identical replicas, ONE gradient exchange
per optimiser step over RCCL/xGMI (replaces the reference's single-process `nn.DataParallel` /
`data_parallel`).

Each model's gradients are gathered into one flat arena by the copy kernel, all-reduced (SUM) with a
single collective, and Adam reads its gradients straight from the arena with the 1/world_size mean
folded into its `gscale` — no copy back, no per-parameter collectives. The backend is whatever
`torch.distributed` was initialised with: "nccl" (= RCCL) on the GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, world).
    T2V_DIST_BACKEND overrides the backend (rehearsing the multi-rank path over gloo on one GPU)."""
    backend = os.environ.get('T2V_DIST_BACKEND', backend)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_device_index())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def local_device_index():
    """LOCAL_RANK, or 0 when T2V_SINGLE_DEVICE=1 (several rehearsal ranks sharing one GPU)."""
    if os.environ.get('T2V_SINGLE_DEVICE', '0') == '1':
        return 0
    return int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))


class GradArena(object):
    """Flat gradient buffer of one model + the per-step exchange."""

    def __init__(self, params, copy_fn=None):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        p0 = self.params[0]
        self.flat = torch.zeros(n, device=p0.device, dtype=p0.dtype)
        self.copy_fn = copy_fn

    def views(self):
        return [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]

    def gather(self):
        """p.grad -> arena slice (missing grads count as zero)."""
        for v, p in zip(self.views(), self.params):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() == v.data_ptr():
                continue                      # the producing kernel already wrote it here (functional.GradSink)
            elif self.copy_fn is not None and p.grad.is_cuda:
                self.copy_fn(p.grad, v)
            else:
                v.copy_(p.grad)

    def all_reduce(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def scatter_as_grads(self):
        """Point every p.grad at its (reduced) arena slice; the optimiser applies `1/world` itself."""
        for v, p in zip(self.views(), self.params):
            p.grad = v


class GradSync(object):
    """The per-step gradient exchange, split so that `GraphedTrainStep` can capture the device-side parts:
    pre(which)      p.grad -> arena slices                (kernel launches only: capturable)
    exchange(which) ONE all_reduce(SUM) of the arena      (collective: stays eager, between graph replays)
    post(which)     p.grad := arena views, Adam gscale    (host bookkeeping only)
    `which` is 'D' or 'G'. Calling the object does all three (eager mode)."""

    def __init__(self, arenas, optimizers, world):
        self.arenas, self.optimizers, self.world = arenas, optimizers, world

    def pre(self, which):
        self.arenas[which].gather()

    def exchange(self, which):
        self.arenas[which].all_reduce()

    def post(self, which):
        self.arenas[which].scatter_as_grads()
        self.optimizers[which].grad_scale = 1.0 / self.world

    def __call__(self, which):
        self.pre(which)
        self.exchange(which)
        self.post(which)


def make_grad_sync(arenas, optimizers, world):
    """arenas / optimizers: dicts keyed 'D' / 'G'."""
    return GradSync(arenas, optimizers, world)
