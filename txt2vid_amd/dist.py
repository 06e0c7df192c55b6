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


def rank_world():
    """(rank, world) of this process; (0, 1) outside a process group."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_seed(seed):
    """Every rank adopts rank 0's seed (the reference draws one with `random.randint` when --seed is absent,
    train/setup.py:8: under one process per GPU each rank would otherwise build a different model)."""
    if rank_world()[1] == 1:
        return seed
    box = [seed]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def sync_replicas(modules, src=0):
    """Broadcast every parameter and buffer of `modules` from rank `src`: the replicas start bit-identical whatever each rank's
    generator state was (random init, a checkpoint only rank 0 could read, ...). Gradient averaging keeps them identical
    afterwards. Returns the number of tensors sent."""
    if rank_world()[1] == 1:
        return 0
    n = 0
    with torch.no_grad():
        for m in modules:
            if m is None:
                continue
            for t in list(m.parameters()) + list(m.buffers()):
                if t.is_contiguous():
                    dist.broadcast(t, src=src)
                else:                                       # (tap-major master copies and the like)
                    c = t.contiguous()
                    dist.broadcast(c, src=src)
                    t.copy_(c)
                n += 1
    return n


def _zeros(n, device, dtype):
    """torch.zeros; fp32 device buffers go through the fill kernel (functional.zeros)."""
    if torch.device(device).type == 'cuda' and dtype == torch.float32:
        from . import functional as TF
        return TF.zeros((n,), device, dtype)
    return torch.zeros(n, device=device, dtype=dtype)


def _is_tap_major_view(v):
    """True when the memory behind a conv-weight-shaped view is laid out [*k][Cout][Cin] (functional.tap_major)."""
    nd = v.dim()
    return nd >= 3 and not v.is_contiguous() and v.permute(*(tuple(range(2, nd)) + (0, 1))).is_contiguous()


class GradArena(object):
    """Flat gradient buffer of one model + the per-step exchange.

    `live_taps` (optional): {parameter: [tap indices]} for convolution weights whose gradient is structurally zero outside
    those kernel taps — the generator's ConvLSTM runs 3x3 kernels on a 1x1 feature map, so only the centre tap of its eight
    [1024,1024,3,3] weights (302 of G's 345 MB) ever receives a gradient. Such parameters sit at the end of the arena and
    only their live taps, packed into a compact buffer, take part in the all-reduce."""

    def __init__(self, params, copy_fn=None, live_taps=None, exchange_dtype=None):
        """`exchange_dtype`: None / torch.float32 = exchange the fp32 arena as it is; torch.bfloat16 (or the environment's
        T2V_GRAD_EXCHANGE=bf16) = the opt-in half-size exchange: the arena is rounded to bf16 into one buffer, ONE all_reduce(SUM)
        runs on it and the sums are widened back into the fp32 arena (SURVEY §8(e) "optionally reduce in bf16"). Every rank
        receives the same bf16 sums, so the replicas stay bit-identical; the gradients carry 8 significant bits."""
        if exchange_dtype is None and os.environ.get('T2V_GRAD_EXCHANGE', '').lower() in ('bf16', 'bfloat16'):
            exchange_dtype = torch.bfloat16
        self.exchange_dtype = exchange_dtype if exchange_dtype in (torch.bfloat16,) else None
        params = [p for p in params if p.requires_grad]
        live_taps = live_taps or {}
        sparse_ids = {id(p) for p in live_taps}
        dense = [p for p in params if id(p) not in sparse_ids]
        sparse = [p for p in params if id(p) in sparse_ids]
        self.params = dense + sparse
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.numel_dense = sum(p.numel() for p in dense)
        p0 = self.params[0]
        self.flat = _zeros(n, p0.device, p0.dtype)
        self.copy_fn = copy_fn
        self.sparse = []                       # (offset in flat, rows = Cout*Cin, taps per row T, live tap list, offset in compact, tap-major?)
        m = 0
        for p in sparse:
            taps = [int(t) for t in live_taps[p]]
            T = 1
            for d in p.shape[2:]:
                T *= int(d)
            rows = p.numel() // T
            self.sparse.append((self.offsets[len(dense) + len(self.sparse)], rows, T, taps, m, not p.is_contiguous()))
            m += rows * len(taps)
        self.compact = _zeros(m, p0.device, p0.dtype) if m else None
        self.half = None
        if self.exchange_dtype is not None:
            self.half_dense = (self.numel_dense + 7) // 8 * 8             # (the compact part starts 16-byte aligned)
            self.half = torch.zeros(self.half_dense + m, device=p0.device, dtype=self.exchange_dtype)

    def views(self):
        """One view per parameter with the parameter's own shape AND strides (tap-major master weights keep their layout: the
        optimiser walks parameter, gradient and moments with the same flat index)."""
        return [self.flat[o:o + p.numel()].view_as(p) if p.is_contiguous() else torch.as_strided(self.flat, p.shape, p.stride(), o)
                for o, p in zip(self.offsets, self.params)]

    def gather(self):
        """p.grad -> arena slice (missing grads count as zero). `copy_fn` is a FLAT copy (dense source -> the destination's
        memory in order), so it only takes pairs whose memory orders agree; a dense gradient for a tap-major slot (a ConvLSTM
        weight whose gradient bypassed the sink: --end2end, T2V_NO_DEFERRED_REDUCE, one-step sequences) goes tap row by tap row."""
        if self.flat.is_cuda:
            from . import functional as TF
            TF.grad_sink_flush()              # pending k-split partial sums belong in the arena before anyone reads it
        for v, p in zip(self.views(), self.params):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() == v.data_ptr() and g.stride() == v.stride():
                continue                      # the producing kernel already wrote it here (functional.GradSink)
            elif self.copy_fn is not None and g.is_cuda and g.is_contiguous() and v.is_contiguous():
                self.copy_fn(g, v)
            elif self.copy_fn is not None and g.is_cuda and g.is_contiguous() and v.dim() >= 3 and _is_tap_major_view(v):
                from . import functional as TF
                T = 1
                for d in v.shape[2:]:
                    T *= int(d)
                rows = v.numel() // T         # arena memory [T][Cout*Cin] <- gradient memory [Cout*Cin][T]: ONE transpose launch
                dst = torch.as_strided(self.flat, (T * rows,), (1,), v.storage_offset())
                TF.check(TF.lib().t2v_permute01(g.data_ptr(), dst.data_ptr(), rows, T, 1, TF._stream()), 't2v_permute01')
            else:
                v.copy_(g)                    # stride-aware (any other layout pair; CPU)

    def _taps(self, pack):
        """live taps: arena -> compact (pack) or compact -> arena (unpack). Tap-major masters ([T][rows] in memory: a live tap is
        one contiguous run — every ConvLSTM weight of the generator) go in ONE multi-job launch per direction on the GPU
        (t2v_multi, 8 runs per launch); anything else takes the strided column copies."""
        if self.flat.is_cuda and self.copy_fn is not None and any(e[5] for e in self.sparse):
            from . import functional_multi as FM
            jobs = []
            for off, rows, T, taps, coff, tap_major in self.sparse:
                if not tap_major:
                    continue
                for j, t in enumerate(taps):
                    a = self.flat[off + t * rows:off + (t + 1) * rows]
                    b = self.compact[coff + j * rows:coff + (j + 1) * rows]
                    src_, dst_ = (a, b) if pack else (b, a)
                    jobs.append(dict(a=src_, out=dst_, n=1, d0=rows, d1=0, d2=rows))       # out[0, 0:rows] = a[0, 0:rows]
            FM._mj(FM.MJ_SLICECOLS, jobs)
            rest = [e for e in self.sparse if not e[5]]
        else:
            rest = self.sparse
        for off, rows, T, taps, coff, tap_major in rest:
            if tap_major:                     # [T][rows] in memory: a live tap is one contiguous run
                for j, t in enumerate(taps):
                    a = self.flat[off + t * rows:off + (t + 1) * rows]
                    b = self.compact[coff + j * rows:coff + (j + 1) * rows]
                    src_, dst_ = (a, b) if pack else (b, a)
                    if self.flat.is_cuda and self.copy_fn is not None:
                        self.copy_fn(src_, dst_)
                    else:
                        dst_.copy_(src_)
                continue
            src = self.flat[off:off + rows * T].view(rows, T)
            dst = self.compact[coff:coff + rows * len(taps)].view(rows, len(taps))
            for j, t in enumerate(taps):
                if self.flat.is_cuda:
                    from . import functional as TF
                    if pack:
                        TF._copy2d(self.flat, off + t, T, self.compact, coff + j, len(taps), rows, 1)
                    else:
                        TF._copy2d(self.compact, coff + j, len(taps), self.flat, off + t, T, rows, 1)
                elif pack:
                    dst[:, j].copy_(src[:, t])
                else:
                    src[:, t].copy_(dst[:, j])

    def _cast(self, src, dst, to_half):
        if src.is_cuda:
            from ._ops import lib, check, _stream
            check(lib().t2v_cast_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), 0 if to_half else 1, _stream()), 't2v_cast_bf16')
        else:
            dst.copy_(src)                                           # (CPU rehearsal over gloo: torch's round-to-nearest-even)

    def _all_reduce_half(self):
        """The opt-in bf16 exchange: [dense | live taps] rounded into one bf16 buffer, ONE collective, widened back."""
        nd, m = self.numel_dense, (self.compact.numel() if self.compact is not None else 0)
        if m:
            self._taps(True)
            self._cast(self.compact, self.half[self.half_dense:], True)
        if nd:
            self._cast(self.flat[:nd], self.half[:nd], True)
        dist.all_reduce(self.half, op=dist.ReduceOp.SUM)
        if nd:
            self._cast(self.half[:nd], self.flat[:nd], False)
        if m:
            self._cast(self.half[self.half_dense:], self.compact, False)
            self._taps(False)

    def all_reduce(self):
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        if self.half is not None:
            self._all_reduce_half()
            return
        if not self.sparse:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            return
        if self.numel_dense:
            dist.all_reduce(self.flat[:self.numel_dense], op=dist.ReduceOp.SUM)
        self._taps(True)
        dist.all_reduce(self.compact, op=dist.ReduceOp.SUM)
        self._taps(False)

    def exchanged_bytes(self):
        return (2 if self.half is not None else 4) * (self.numel_dense + (self.compact.numel() if self.compact is not None else 0))

    def scatter_as_grads(self):
        """Point every p.grad at its (reduced) arena slice; the optimiser applies `1/world` itself."""
        for v, p in zip(self.views(), self.params):
            p.grad = v


def model_arena(modules, copy_fn=None, exchange_dtype=None):
    """GradArena over the parameters of one module (or a list of modules); modules that know which kernel taps of their
    convolution weights can ever receive a gradient (`structurally_live_taps()`) have only those exchanged."""
    if not isinstance(modules, (list, tuple)):
        modules = [modules]
    params, live = [], {}
    for m in modules:
        params += list(m.parameters())
        if hasattr(m, 'structurally_live_taps'):
            live.update(m.structurally_live_taps())
    return GradArena(params, copy_fn, live_taps=live, exchange_dtype=exchange_dtype)


class GradSync(object):
    """The per-step gradient exchange, split so that `GraphedTrainStep` can capture the device-side parts:
    pre(which)      p.grad -> arena slices                (kernel launches only: capturable)
    exchange(which) ONE all_reduce(SUM) of the arena      (collective: stays eager, between graph replays)
    post(which)     p.grad := arena views, Adam gscale    (host bookkeeping only)
    `which` is 'D' or 'G'. Calling the object does all three (eager mode)."""

    def __init__(self, arenas, optimizers, world):
        self.arenas, self.optimizers, self.world = arenas, optimizers, world
        self.timing = None                  # list of (which, start event, end event) while `time_exchanges(True)`

    def pre(self, which):
        self.arenas[which].gather()

    def time_exchanges(self, on=True):
        """Bracket every exchange with a pair of events on the stream that issues it (bench.py: `allreduce_ms_per_step`). The
        collective runs on the backend's own stream between two waits on this one, so the pair spans exactly what the iteration
        waits for."""
        self.timing = [] if on else None

    def exchange_ms(self):
        """Total milliseconds of the exchanges recorded since `time_exchanges(True)`, per model: {'D': ms, 'G': ms, 'n': exchanges}.
        (Host arenas — the gloo rehearsal on CPU tensors — are timed with the host clock around the blocking collective.)"""
        out = {'D': 0.0, 'G': 0.0, 'n': 0}
        for which, e0, e1 in self.timing or ():
            if e1 is None:
                ms = e0
            else:
                e1.synchronize()
                ms = e0.elapsed_time(e1)
            out[which] = out.get(which, 0.0) + ms
            out['n'] += 1
        return out

    def exchange(self, which):
        flat = self.arenas[which].flat
        if self.timing is not None and flat.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.arenas[which].all_reduce()
            e1.record()
            self.timing.append((which, e0, e1))
            return
        if self.timing is not None:
            import time
            t0 = time.perf_counter()
            self.arenas[which].all_reduce()
            self.timing.append((which, (time.perf_counter() - t0) * 1e3, None))
            return
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
