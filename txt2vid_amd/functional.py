"""Autograd surface over the C-ABI kernels (include/t2v_hip.h).

Every op on the TGANv2 hot path is a `torch.autograd.Function` whose forward AND backward launch
hand-written gfx950 kernels through ctypes on torch's current HIP stream. torch is used only for
device memory (torch.empty), the stream and the autograd graph bookkeeping.

The discriminator ops are *closed under differentiation*: the backward of each Function is itself
written with Functions of this module, so `torch.autograd.grad(..., create_graph=True)` — the
gradient penalty of txt2vid/gan/losses.py:178 — yields a graph whose own backward again runs only
these kernels (conv fwd <-> dgrad <-> wgrad form a closed triple).
"""
import contextlib
import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import lib, check, ConvGeom, ConvGroup, PackJob, WgradSrc, WgradDest, WGRAD_MAX_SRC, MAX_TAPS, MAX_GROUPS

FLAG_BIAS, FLAG_RELU_IN, FLAG_ACCUM, FLAG_MASK_OUT, FLAG_ACCUM_BIAS, FLAG_BF16 = 1, 2, 4, 8, 16, 32

# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------

from ._ops import _stream, _p, _c          # noqa: E402,F401


def tap_major(w):
    """The same conv weight [Cout,Cin,*k] with its MEMORY laid out [*k][Cout][Cin] (a permuted view of a dense buffer)."""
    nd = w.dim()
    fwd = tuple(range(2, nd)) + (0, 1)
    back = tuple(range(nd - 2, nd)) + tuple(range(nd - 2))
    return w.permute(*fwd).contiguous().permute(*back)


def is_tap_major(w):
    """True for the layout `tap_major` produces (False for dense weights; anything else is rejected)."""
    if w.is_contiguous():
        return False
    nd = w.dim()
    if nd >= 3 and w.permute(*(tuple(range(2, nd)) + (0, 1))).is_contiguous():
        return True
    raise ValueError('conv weights are dense [Cout,Cin,*k] or tap-major; got strides %s for shape %s' % (w.stride(), tuple(w.shape)))


def tap_rows(t):
    """[T, Cout*Cin] view of a tap-major tensor (each kernel tap one contiguous row)."""
    nd = t.dim()
    v = t.permute(*(tuple(range(2, nd)) + (0, 1)))
    return v.reshape(-1, t.shape[0] * t.shape[1])


_param_grads_enabled = True
WEIGHT_EPOCH = 0     # bumped by the optimiser after each in-place update (invalidates packed weights)


@contextlib.contextmanager
def input_grads_only():
    """Inside this context the backward of conv/linear ops computes only the data gradient.
    Used around the `autograd.grad(outputs, inputs=[x_hat])` sweep of the gradient penalty, where
    torch would otherwise make every custom Function produce (and record a graph for) weight
    gradients nobody asked for (`ctx.needs_input_grad` is static for Python Functions)."""
    global _param_grads_enabled
    old = _param_grads_enabled
    _param_grads_enabled = False
    try:
        yield
    finally:
        _param_grads_enabled = old


_NO_VEC_REDUCE = bool(os.environ.get('T2V_NO_VEC_REDUCE'))       # developer A/B switch: scalar slab loads in the batched reduce


class _PendingDest(object):
    __slots__ = ('wid', 'bid', 'dw', 'dbias', 'CoCi', 'T', 'Cout', 'accum', 'accum_bias', 'srcs', 'keep', 'tap_major')


class GradSink(object):
    """Weight / bias gradients written straight into a flat per-model arena (`dist.GradArena`) by the kernels that
    produce them: the first producer of a step stores, later producers of the same parameter accumulate in place
    (T2V_CONV_ACCUM) and hand autograd nothing — no per-parameter sum kernels, and the data-parallel exchange finds
    the gradients already in its buffer. Only used while autograd is not recording (the final `loss.backward()`);
    `reset()` belongs wherever the model's gradients are cleared.

    Convolution weight gradients are DEFERRED: each layer's weight-gradient launch leaves its k-split partial sums in its
    slab (`t2v_conv_wgrad_grouped_partial`) and `flush()` — called once after the backward pass — sums every pending slab of
    every parameter in ONE launch (`t2v_wgrad_reduce_multi`) instead of one ~10 us reduce launch per layer. A producer
    that writes a parameter's slot directly while slabs are pending for it flushes that parameter first (ordering)."""

    def __init__(self, arenas):
        self.slots, self.count = {}, {}
        for arena in arenas:
            for off, p in zip(arena.offsets, arena.params):
                self.slots[id(p)] = (arena, off, p.numel())
        self.defer = os.environ.get('T2V_NO_DEFERRED_REDUCE') is None
        self.pending, self.pending_bias = {}, {}          # id(weight) -> _PendingDest; id(bias) -> id(weight)
        # Destination tables live in a fixed device buffer, uploaded from a fixed pinned buffer (both allocated HERE, outside
        # any graph capture): a table is uploaded once per distinct content (slot per signature) — a copy from pinned memory
        # is also a legal graph node should a capture meet a new table.
        self.tables = {}                                  # table bytes -> slot
        self.slot_bytes, self.nslots = 65536, 48
        dev = arenas[0].flat.device if arenas else None
        self.tab_pin = self.tab_dev = None
        if dev is not None and dev.type == 'cuda' and self.defer:
            self.tab_pin = torch.empty(self.slot_bytes * self.nslots, dtype=torch.uint8).pin_memory()
            self.tab_dev = torch.empty(self.slot_bytes * self.nslots, dtype=torch.uint8, device=dev)
        else:
            self.defer = False
        # k-split slabs come from one bump-allocated workspace sized by the first iteration: every later iteration — eager,
        # captured or replayed — hands each launch the same address, so the destination tables repeat and are uploaded once
        self.ws, self.ws_off, self.ws_used = None, 0, 0
        # captured HIP graphs bake in the slab addresses and the table slots they met: while one is alive (`frozen` > 0, set by
        # GraphedTrainStep) the workspace must not be re-allocated and the table slots must not be recycled
        self.frozen = 0

    def reset(self):
        self.flush()
        self.count.clear()
        grow = self.ws_used > (self.ws.numel() if self.ws is not None else 0)
        if grow and self.frozen:
            raise RuntimeError('the gradient sink would have to grow its slab workspace (%d -> %d floats) while a captured HIP graph '
                               'still reads the current one: release that graph first (GraphedTrainStep.release)'
                               % (self.ws.numel() if self.ws is not None else 0, self.ws_used))
        if grow and not torch.cuda.is_current_stream_capturing():
            self.ws = None                                # (release before growing)
            self.ws = torch.empty(self.ws_used + (self.ws_used >> 3), dtype=torch.float32, device=self.tab_dev.device)
            torch.cuda.synchronize()                       # (first iterations only) pending table uploads — on any stream — are done with their slots
            self.tables.clear()
        self.ws_off = self.ws_used = 0

    def alloc_slab(self, n, device):
        """`n` floats for one launch's partial sums: from the workspace when it has room, else a tensor of its own (first
        iteration, or a pass that needs more than any before)."""
        n = (int(n) + 63) & ~63
        self.ws_used += n
        if self.ws is not None and self.ws_off + n <= self.ws.numel():
            v = self.ws[self.ws_off:self.ws_off + n]
            self.ws_off += n
            return v
        return torch.empty((n,), device=device, dtype=torch.float32)

    def take(self, param, deferred=False):
        slot = self.slots.get(id(param))
        if slot is None:
            return None
        if not deferred and self.pending:
            wid = id(param) if id(param) in self.pending else self.pending_bias.get(id(param))
            if wid is not None:
                self._flush([self.pending[wid]])
        arena, off, n = slot
        c = self.count.get(id(param), 0)
        self.count[id(param)] = c + 1
        return arena.flat[off:off + n], c > 0

    # ---- deferred weight gradients
    def add_partial(self, wbase, dw, wacc, b, dbias, bacc, src, keep, T, Cout, CoCi, tap_major=False):
        """Register one weight-gradient launch's slab (`src`: a filled WgradSrc) for parameter `wbase` (+ bias `b`)."""
        d = self.pending.get(id(wbase))
        if d is not None and (len(d.srcs) >= WGRAD_MAX_SRC or (d.bid is not None and b is not None and d.bid != id(b))):
            self._flush([d])
            # the flushed table wrote the weight slot; the bias slot only if it carried THIS bias (else the incoming producer may
            # still be the step's first writer of its bias and must overwrite, not add to last step's contents)
            wacc, bacc = True, bacc or (d.bid is not None and b is not None and d.bid == id(b))
            d = None
        if d is None:
            d = _PendingDest()
            d.wid, d.bid, d.dw, d.dbias = id(wbase), None, dw.data_ptr(), 0
            d.CoCi, d.T, d.Cout, d.accum, d.accum_bias, d.srcs, d.keep = CoCi, T, Cout, 1 if wacc else 0, 0, [], []
            d.tap_major = 1 if tap_major else 0
            self.pending[id(wbase)] = d
        if b is not None and d.bid is None:
            d.bid, d.dbias, d.accum_bias = id(b), dbias.data_ptr(), 1 if bacc else 0
            self.pending_bias[id(b)] = id(wbase)
        d.srcs.append(src)
        d.keep.append(keep)

    def flush(self):
        if self.pending:
            self._flush(list(self.pending.values()))

    def _flush(self, dests):
        _side.join()                                       # (slabs written on the side stream)
        arr = (WgradDest * len(dests))()
        blocks = 0
        sig = []
        for a, d in zip(arr, dests):
            a.dw, a.dbias, a.CoCi, a.T, a.Cout = d.dw, d.dbias or None, d.CoCi, d.T, d.Cout
            a.nsrc, a.accum, a.accum_bias, a.tap_major = len(d.srcs), d.accum, d.accum_bias, d.tap_major
            smax = max(sr.S for sr in d.srcs)
            vec = not _NO_VEC_REDUCE and d.CoCi % 4 == 0 and d.CoCi >= 1024 and all(sr.slab % 16 == 0 and sr.tap_stride % 4 == 0 and sr.split_stride % 4 == 0
                                                             for sr in d.srcs)
            if vec:                  # 16-byte slab loads, 256 pairs per workgroup; same split between the two forms (and same
                a.kind = 3 if (d.CoCi <= 16384 and smax >= 16) else 2          # summation order) as the scalar kinds 1 / 0
                a.nblocks = ((d.CoCi + 255) // 256) * (d.T if a.kind == 3 else 1)
            else:
                a.kind = 1 if (d.CoCi <= 16384 and smax >= 16) else 0
                a.nblocks = ((d.CoCi + 63) // 64) * (d.T if a.kind else 1)
            a.block_begin = blocks
            blocks += a.nblocks
            for k, sr in enumerate(d.srcs):
                C.memmove(C.byref(a.src[k]), C.byref(sr), C.sizeof(WgradSrc))
        raw = bytes(arr)
        slot = self.tables.get(raw)
        if slot is None:
            assert C.sizeof(WgradDest) == lib().t2v_wgrad_dest_bytes()
            if len(raw) > self.slot_bytes:
                raise RuntimeError('weight-gradient destination table of %d bytes' % len(raw))
            if len(self.tables) >= self.nslots:
                if torch.cuda.is_current_stream_capturing() or self.frozen:
                    raise RuntimeError('out of destination-table slots %s' % ('during graph capture' if not self.frozen else
                                                                               'while a captured HIP graph still reads them'))
                torch.cuda.synchronize()                   # (eager, pointers drifting: earlier uploads, on any stream, are done with their slots)
                self.tables.clear()
            slot = len(self.tables)
            lo = slot * self.slot_bytes
            self.tab_pin[lo:lo + len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))     # host memcpy
            self.tab_dev[lo:lo + len(raw)].copy_(self.tab_pin[lo:lo + len(raw)], non_blocking=True)
            self.tables[raw] = slot
        check(lib().t2v_wgrad_reduce_multi(C.c_void_p(self.tab_dev.data_ptr() + slot * self.slot_bytes), len(dests), blocks, _stream()),
              't2v_wgrad_reduce_multi')
        for d in dests:
            self.pending.pop(d.wid, None)
            if d.bid is not None:
                self.pending_bias.pop(d.bid, None)


_grad_sink = None


class _SideLane:
    """A second HIP stream for the deferred weight-gradient launches of a backward pass. A layer's weight gradient depends on
    dL/dy and the saved input only, and nothing reads it before the sink's flush: launched on the main stream it sits IN the
    data-gradient chain, and on the deep blocks (a few hundred voxels, a handful of workgroups walking K of several thousand)
    both kernels leave most of the chip idle. Forked after the producer of dL/dy (event on the main stream) and joined in
    `GradSink._flush`, the same launches overlap the chain; under capture the fork / join become graph edges."""

    def __init__(self):
        self.enabled = os.environ.get('T2V_WGRAD_SIDE') is not None
        self.stream, self.used, self.keep = None, False, []

    def fork(self, *tensor_lists):
        """The stream handle for one launch that may overlap what follows on the current stream; `tensor_lists` are its
        operands, kept alive until the join (they were allocated on, and would be recycled by, the main stream)."""
        if not self.enabled:
            return _stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        self.used = True
        for ts in tensor_lists:
            self.keep.extend(ts)
        return C.c_void_p(self.stream.cuda_stream)

    def join(self):
        if self.used:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.used = False
            self.keep.clear()


_side = _SideLane()


def set_grad_sink(sink):
    global _grad_sink
    old, _grad_sink = _grad_sink, sink
    if old is not None and old is not sink:
        old.flush()
    return old


def grad_sink_reset():
    if _grad_sink is not None:
        _grad_sink.reset()


def grad_sink_flush():
    """Sum every pending weight-gradient slab into its parameter's slot (one launch). Call after a backward pass, before
    anything reads the gradients (optimiser step, gradient exchange)."""
    if _grad_sink is not None:
        _grad_sink.flush()


def _to_sink_wb(w, b, xs, gys, relu_in, even_frames=False):
    """Weight AND bias gradient of one (grouped) convolution into their sink slots with ONE weight-gradient launch
    (`t2v_conv_wgrad_grouped_bias`: the 3-tap-row kernel sums the dL/dy tiles it stages anyway). Returns
    (handled, gw, gb) like `_to_sink`."""
    if _grad_sink is None or w is None or b is None or torch.is_grad_enabled():
        return False, None, None
    wbase = w._base if w._base is not None else w
    if id(wbase) not in _grad_sink.slots or id(b) not in _grad_sink.slots:
        return False, None, None
    deferred = _grad_sink.defer and _wgrad_bias_fused(xs, tuple(w.shape))
    wflat, wacc = _grad_sink.take(wbase, deferred)
    bflat, bacc = _grad_sink.take(b, deferred)
    if deferred:
        conv_group_wgrad_partial(xs, gys, tuple(w.shape), relu_in, _grad_sink, wbase, wflat.view(w.shape), wacc, b, bflat, bacc,
                                 even_frames=even_frames)
    else:
        conv_group_wgrad_raw(xs, gys, tuple(w.shape), relu_in, out=wflat.view(w.shape), accum=wacc, dbias=bflat, accum_bias=bacc,
                             even_frames=even_frames)
    return True, (None if wacc else wflat.view(w.shape)), (None if bacc else bflat.view(b.shape))


def _to_sink_w(w, xs, gys, relu_in, even_frames=False):
    """Weight gradient of one (grouped) convolution into its sink slot, deferred when the sink batches its reductions.
    Returns (handled, gw) like `_to_sink`."""
    if _grad_sink is None or w is None or torch.is_grad_enabled():
        return False, None
    wbase = w._base if w._base is not None else w
    if id(wbase) not in _grad_sink.slots:
        return False, None
    wflat, wacc = _grad_sink.take(wbase, _grad_sink.defer)
    if _grad_sink.defer:
        conv_group_wgrad_partial(xs, gys, tuple(w.shape), relu_in, _grad_sink, wbase, wflat.view(w.shape), wacc, None, None, False,
                                 even_frames=even_frames)
    else:
        conv_group_wgrad_raw(xs, gys, tuple(w.shape), relu_in, out=wflat.view(w.shape), accum=wacc, even_frames=even_frames)
    return True, (None if wacc else wflat.view(w.shape))


def _to_sink(param_like, compute):
    """If `param_like` (a parameter or a view of one) has a sink slot: run compute(out, accum) into it and return
    (True, grad for autograd — a fresh view the first time, None when accumulated). Else (False, None)."""
    if _grad_sink is None or param_like is None or torch.is_grad_enabled():
        return False, None
    base = param_like._base if param_like._base is not None else param_like
    slot = _grad_sink.take(base)
    if slot is None:
        return False, None
    flat, accum = slot
    compute(flat.view(param_like.shape), accum)
    return True, (None if accum else flat.view(param_like.shape))


def bump_weight_epoch(params=None):
    """Tell the packed-weight caches that parameters were rewritten behind autograd's back (raw-pointer kernels do
    not bump `_version`): the given ones, or — with no argument — every parameter."""
    global WEIGHT_EPOCH
    if params is None:
        WEIGHT_EPOCH += 1
        return
    for p in params:
        p._t2v_epoch = getattr(p, '_t2v_epoch', 0) + 1


def _wtag(base, ptr=None):
    return (base._version, WEIGHT_EPOCH, getattr(base, '_t2v_epoch', 0), base.data_ptr() if ptr is None else ptr)


# ------------------------------------------------------------------------------------------------
# convolution geometry + packed-weight cache
# ------------------------------------------------------------------------------------------------

class _Geom(object):
    __slots__ = ('cg', 'taps', 'taps_c', 'T', 'mask', 'kshape', 'ws_floats')


_geom_cache = {}


def conv_geom(N, Cin, D, H, W, Cout, kD, kH, kW):
    key = (N, Cin, D, H, W, Cout, kD, kH, kW)
    g = _geom_cache.get(key)
    if g is not None:
        return g
    for k in (kD, kH, kW):
        if k not in (1, 3):
            raise ValueError('only 1- and 3-wide stride-1 same-padded kernels are on the hot path')
    cg = ConvGeom()
    cg.N, cg.Cin, cg.D, cg.H, cg.W, cg.Cout = N, Cin, D, H, W, Cout
    taps = []
    for a in range(kD):
        for b in range(kH):
            for c in range(kW):
                dz, dy, dx = a - kD // 2, b - kH // 2, c - kW // 2
                if (D == 1 and dz != 0) or (H == 1 and dy != 0) or (W == 1 and dx != 0):
                    continue                          # this tap only ever multiplies padding
                j = len(taps)
                cg.dz[j], cg.dy[j], cg.dx[j] = dz, dy, dx
                taps.append((a * kH + b) * kW + c)
    cg.ntaps = len(taps)
    g = _Geom()
    g.cg, g.taps, g.T, g.kshape = cg, taps, kD * kH * kW, (kD, kH, kW)
    g.taps_c = (C.c_int32 * len(taps))(*taps)
    g.mask = sum(1 << t for t in taps)
    g.ws_floats = None
    _geom_cache[key] = g
    return g


_pack_cache = {}      # id(base parameter) -> {(shape5, tap mask, mode): [weakref, tag, buffer, geom]}


def packed_weight(w5, geom, mode):
    """wp[ntaps][Cin][Cout] (mode 0) / wp[ntaps][Cout][Cin] mirrored (mode 1). For parameters the packed
    buffer is PERSISTENT (one per (parameter, tap set, mode)) and is refreshed in place — lazily here when
    the parameter changed, eagerly by `repack_params` right after an optimiser step — so that a captured
    HIP graph always reads the same addresses."""
    Cout, Cin = w5.shape[0], w5.shape[1]
    base = w5._base if w5._base is not None else w5          # 2-D convs / Linear arrive as 5-D views
    cacheable = isinstance(base, torch.nn.Parameter)
    if not cacheable:
        wp = torch.empty((len(geom.taps), Cin * Cout), device=w5.device, dtype=torch.float32)
        check(lib().t2v_pack_weight(_p(w5), _p(wp), Cout, Cin, geom.T, geom.taps_c, len(geom.taps), mode, _stream()),
              't2v_pack_weight')
        return wp
    key = (tuple(w5.shape), geom.mask, mode)
    tag = _wtag(base, w5.data_ptr())
    ent = _pack_cache.setdefault(id(base), {})
    hit = ent.get(key)
    if hit is not None and hit[0]() is base:
        if hit[1] != tag:
            check(lib().t2v_pack_weight(_p(w5), _p(hit[2]), Cout, Cin, geom.T, geom.taps_c, len(geom.taps), mode, _stream()),
                  't2v_pack_weight')
            hit[1] = tag
        return hit[2]
    wp = torch.empty((len(geom.taps), Cin * Cout), device=w5.device, dtype=torch.float32)
    check(lib().t2v_pack_weight(_p(w5), _p(wp), Cout, Cin, geom.T, geom.taps_c, len(geom.taps), mode, _stream()),
          't2v_pack_weight')
    ent[key] = [weakref.ref(base), tag, wp, geom]
    return wp


CONV_PRECISION = os.environ.get('T2V_CONV_PRECISION', 'fp32')     # 'bf16': forward / data-gradient GEMMs on bf16 MFMA (fp32 storage
#                                                                   and accumulation); opt-in, see set_conv_precision


def set_conv_precision(p):
    """'fp32' (default: what the parity tests and the benchmark metric use) or 'bf16' (BASELINE configs 2-4)."""
    global CONV_PRECISION
    if p not in ('fp32', 'bf16'):
        raise ValueError(p)
    old, CONV_PRECISION = CONV_PRECISION, p
    return old


def packed_weight_bf16(w5, ts, mode):
    """bf16 wpb[ntaps][rows][K] (K contiguous) for the bf16-compute GEMM; persistent per (parameter, tap set, mode) and
    refreshed lazily when the parameter changed (inside the captured graph under replay)."""
    Cout, Cin = w5.shape[0], w5.shape[1]
    base = w5._base if w5._base is not None else w5

    def pack(dst):
        check(lib().t2v_pack_weight_bf16(_p(w5), _p(dst), Cout, Cin, ts.T, ts.taps_c, len(ts.taps), mode, _stream()),
              't2v_pack_weight_bf16')
    if not isinstance(base, torch.nn.Parameter):
        wp = torch.empty((len(ts.taps), Cin * Cout), device=w5.device, dtype=torch.bfloat16)
        pack(wp)
        return wp
    key = ('bf16', tuple(w5.shape), ts.mask, mode)
    tag = _wtag(base, w5.data_ptr())
    ent = _pack_cache.setdefault(id(base), {})
    hit = ent.get(key)
    if hit is not None and hit[0]() is base:
        if hit[1] != tag:
            pack(hit[2])
            hit[1] = tag
        return hit[2]
    wp = torch.empty((len(ts.taps), Cin * Cout), device=w5.device, dtype=torch.bfloat16)
    pack(wp)
    ent[key] = [weakref.ref(base), tag, wp, ts]
    return wp


_repack_tables = {}


def _pack_job_of(p, key, hit):
    """(src, dst, Cout, Cin, T, taps, mode, dst_rows, dst_cols, row_off, col_off) of one cached packed variant."""
    if callable(hit[3]):
        return hit[4]                    # member of a fused matrix: its job tuple was recorded at creation
    if key[0] == 'bf16':                 # bf16-compute layouts: pack_multi modes 2 (forward) / 3 (data gradient)
        _, shape5, _, mode = key
        ts = hit[3]
        Cout, Cin = shape5[0], shape5[1]
        src = p if tuple(p.shape) == tuple(shape5) else p.view(shape5)
        return (src, hit[2], Cout, Cin, ts.T, list(ts.taps), 2 + mode, (Cin if mode else Cout), (Cout if mode else Cin), 0, 0)
    shape5, _, mode = key
    geom = hit[3]
    Cout, Cin = shape5[0], shape5[1]
    return (p, hit[2], Cout, Cin, geom.T, list(geom.taps), mode, (Cout if mode else Cin), (Cin if mode else Cout), 0, 0)


def repack_params(params):
    """Refresh every packed variant of these parameters in place, in ONE launch (`t2v_pack_multi`) driven by a
    device-resident job table that is rebuilt only when the set of cached variants changes. Called by the
    optimiser right after its in-place update (inside the captured graph when graphs are in use)."""
    params = list(params)
    jobs = []
    for p in params:
        ent = _pack_cache.get(id(p))
        if not ent:
            continue
        for key, hit in ent.items():
            if hit[0]() is p:
                jobs.append((p, key, hit))
    if not jobs:
        return
    sig = tuple((id(p), key, hit[2].data_ptr(), p.data_ptr()) for p, key, hit in jobs)
    tkey = tuple(id(p) for p in params)
    tab = _repack_tables.get(tkey)
    if tab is None or tab[0] != sig:
        assert C.sizeof(PackJob) == lib().t2v_pack_job_bytes()
        arr = (PackJob * len(jobs))()
        blocks = 0
        for a, (p, key, hit) in zip(arr, jobs):
            src, dst, Cout, Cin, T, taps, mode, rows, cols, ro, co = _pack_job_of(p, key, hit)
            a.src, a.dst = src.data_ptr(), dst.data_ptr()
            a.Cout, a.Cin, a.T, a.ntaps, a.mode = Cout, Cin, T, len(taps), mode
            a.dst_rows, a.dst_cols, a.row_off, a.col_off = rows, cols, ro, co
            a.bx, a.by = (Cin + 31) // 32, (Cout + 31) // 32
            a.block_begin = blocks
            for j, t in enumerate(taps):
                a.taps[j] = t
            blocks += a.bx * a.by * len(taps)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        dev = host.to(params[0].device)
        tab = (sig, dev, len(jobs), blocks)
        _repack_tables[tkey] = tab
    check(lib().t2v_pack_multi(_p(tab[1]), tab[2], tab[3], _stream()), 't2v_pack_multi')
    for p, key, hit in jobs:
        hit[1] = _wtag(p)
    for fkey, fh in _fused_cache.items():
        if any(r() is p for r in fh[0] for p in params):
            fh[1] = tuple(_wtag(w()) for w in fh[0] if w() is not None)      # members are whole-parameter views: same pointer


_fused_cache = {}


def packed_fused(ws, ts, mode):
    """ONE packed matrix for several same-shape weights applied to the same input (mode 0: outputs side by
    side -> wp[tap][Cin][n*Cout]) or whose data gradients add up (mode 1: wp[tap][n*Cout][Cin]).
    Persistent and refreshed per member exactly like `packed_weight` (the members usually arrive as 5-D views of the
    parameters: the cache is keyed by the parameters behind them)."""
    n = len(ws)
    Cout, Cin = ws[0].shape[0], ws[0].shape[1]
    bases = [w._base if w._base is not None else w for w in ws]
    cacheable = all(isinstance(b_, torch.nn.Parameter) for b_ in bases)
    key = (tuple(id(b_) for b_ in bases), tuple(ws[0].shape), ts.mask, mode)
    hit = _fused_cache.get(key) if cacheable else None
    if hit is not None and all(r() is b_ for r, b_ in zip(hit[0], bases)):
        tag = tuple(_wtag(b_, w.data_ptr()) for b_, w in zip(bases, ws))
        if hit[1] != tag:
            for i, w in enumerate(ws):
                hit[3][i](w)
            hit[1] = tag
        return hit[2]
    rows, cols = (Cin, n * Cout) if mode == 0 else (n * Cout, Cin)
    wp = torch.empty((len(ts.taps), rows * cols), device=ws[0].device, dtype=torch.float32)
    refresh = []
    for i, w in enumerate(ws):
        ro, co = (0, i * Cout) if mode == 0 else (i * Cout, 0)

        tm = 8 if is_tap_major(w) else 0         # (mode | 8: the source is stored tap-major)

        def fn(wi, ro=ro, co=co, tm=tm):
            check(lib().t2v_pack_weight_into(_p(wi), _p(wp), Cout, Cin, ts.T, ts.taps_c, len(ts.taps), mode | tm, rows, cols,
                                             ro, co, _stream()), 't2v_pack_weight_into')
        refresh.append(fn)
        fn(w)
        if cacheable:                      # lets `repack_params` refresh this member in its one multi-tensor launch
            _pack_cache.setdefault(id(bases[i]), {})[('fused', key, i)] = [weakref.ref(bases[i]), _wtag(bases[i], w.data_ptr()), wp, fn,
                                                                           (w, wp, Cout, Cin, ts.T, list(ts.taps), mode | tm, rows, cols, ro, co)]
    if cacheable:
        _fused_cache[key] = [[weakref.ref(b_) for b_ in bases], tuple(_wtag(b_, w.data_ptr()) for b_, w in zip(bases, ws)), wp, refresh]
    return wp


def conv_packed_raw(x5, wp, cin, cout, k, bias=None, out=None, accum=False):
    """Convolution with an explicitly packed weight (fused gates): wp holds the taps of this geometry in order."""
    x5 = _c(x5)
    g = conv_geom(x5.shape[0], cin, x5.shape[2], x5.shape[3], x5.shape[4], cout, k[0], k[1], k[2])
    y = out if out is not None else torch.empty((x5.shape[0], cout) + tuple(x5.shape[2:]), device=x5.device, dtype=torch.float32)
    flags = (FLAG_BIAS if bias is not None else 0) | (FLAG_ACCUM if accum else 0)
    check(lib().t2v_conv_fwd(_p(x5), _p(wp), _p(bias), _p(y), _p(_conv_ws(g, x5.device)), C.byref(g.cg), flags, _stream()),
          't2v_conv_fwd')
    return y, g


def _as5(t):
    """view [N,C] / [N,C,H,W] / [N,C,D,H,W] as 5-D."""
    if t.dim() == 5:
        return t
    if t.dim() == 4:
        return t.unsqueeze(2)
    if t.dim() == 2:
        return t.view(t.size(0), t.size(1), 1, 1, 1)
    raise ValueError('expected a 2-, 4- or 5-D tensor')


def _geom_for(x5, w5):
    return conv_geom(x5.shape[0], w5.shape[1], x5.shape[2], x5.shape[3], x5.shape[4], w5.shape[0],
                     w5.shape[2], w5.shape[3], w5.shape[4])


def _conv_ws(g, device):
    """split-K workspace for this geometry (None when the launch plan does not split)."""
    n = getattr(g, 'ws_floats', None)
    if n is None:
        n = int(lib().t2v_conv_fwd_ws_floats(C.byref(g.cg)))
        if n < 0:
            raise RuntimeError('bad conv geometry')
        g.ws_floats = n
    return torch.empty((n,), device=device, dtype=torch.float32) if n > 0 else None


def conv_fwd_raw(x5, w5, bias=None, relu_in=False, out=None, accum=False):
    if CONV_PRECISION == 'bf16':
        return conv_group_raw([x5], w5, bias, relu_in, 0, outs=None if out is None else [out], accum=accum)[0]
    x5, w5 = _c(x5), _c(w5)
    g = _geom_for(x5, w5)
    wp = packed_weight(w5, g, 0)
    y = out if out is not None else torch.empty((x5.shape[0], w5.shape[0]) + tuple(x5.shape[2:]), device=x5.device,
                                                dtype=torch.float32)
    flags = (FLAG_BIAS if bias is not None else 0) | (FLAG_RELU_IN if relu_in else 0) | (FLAG_ACCUM if accum else 0)
    check(lib().t2v_conv_fwd(_p(x5), _p(wp), _p(bias), _p(y), _p(_conv_ws(g, x5.device)), C.byref(g.cg), flags, _stream()),
          't2v_conv_fwd')
    return y


def conv_dgrad_raw(gy5, w5, out=None, accum=False):
    """gx[N,Cin,...] = sum_taps,co gy * w (mirrored): the forward kernel on the mode-1 packed weight."""
    if CONV_PRECISION == 'bf16':
        return conv_group_raw([gy5], w5, None, False, 1, outs=None if out is None else [out], accum=accum)[0]
    gy5, w5 = _c(gy5), _c(w5)
    Cout, Cin = w5.shape[0], w5.shape[1]
    # geometry of the transposed problem: input channels = Cout, output channels = Cin
    gt = conv_geom(gy5.shape[0], Cout, gy5.shape[2], gy5.shape[3], gy5.shape[4], Cin, w5.shape[2], w5.shape[3], w5.shape[4])
    wp = packed_weight(w5, gt, 1)
    gx = out if out is not None else torch.empty((gy5.shape[0], Cin) + tuple(gy5.shape[2:]), device=gy5.device,
                                                 dtype=torch.float32)
    check(lib().t2v_conv_fwd(_p(gy5), _p(wp), None, _p(gx), _p(_conv_ws(gt, gy5.device)), C.byref(gt.cg),
                             FLAG_ACCUM if accum else 0, _stream()), 't2v_conv_fwd(dgrad)')
    return gx


def conv_wgrad_raw(x5, gy5, wshape, relu_in=False, out=None, accum=False):
    if CONV_PRECISION == 'bf16':
        return conv_group_wgrad_raw([x5], [gy5], wshape, relu_in, out=out, accum=accum)
    x5, gy5 = _c(x5), _c(gy5)
    g = conv_geom(x5.shape[0], wshape[1], x5.shape[2], x5.shape[3], x5.shape[4], wshape[0], wshape[2], wshape[3], wshape[4])
    n = lib().t2v_conv_wgrad_slab_floats(C.byref(g.cg), g.T)
    if n <= 0:
        raise RuntimeError('bad wgrad geometry')
    slab = torch.empty((n,), device=x5.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty(tuple(wshape), device=x5.device, dtype=torch.float32)
    check(lib().t2v_conv_wgrad(_p(x5), _p(gy5), _p(dw), _p(slab), C.byref(g.cg), g.taps_c, g.T,
                               (FLAG_RELU_IN if relu_in else 0) | (FLAG_ACCUM if accum else 0), _stream()), 't2v_conv_wgrad')
    return dw


def channel_sum_raw(t5, out=None, accum=False):
    t5 = _c(t5)
    N, Cc = t5.shape[0], t5.shape[1]
    S = t5.numel() // (N * Cc)
    if out is None:
        out = torch.empty((Cc,), device=t5.device, dtype=torch.float32)
    nws = int(lib().t2v_channel_sum_ws_floats(N, Cc, S))
    ws = torch.empty((nws,), device=t5.device, dtype=torch.float32) if nws > 0 else None
    check(lib().t2v_channel_sum(_p(t5), _p(out), _p(ws), N, Cc, S, 1 if accum else 0, _stream()), 't2v_channel_sum')
    return out


class Conv(Function):
    """y = conv(x, w) + b  (stride 1, same padding; 5-D tensors). nn.Conv3d/Conv2d/Linear forward."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.bias = b is not None, b
        return conv_fwd_raw(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ConvDgrad.apply(gy, w)
        if _param_grads_enabled:
            both = False
            if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
                both, gw, gb = _to_sink_wb(w, ctx.bias, [x], [gy], False)
            if not both:
                if ctx.needs_input_grad[1]:
                    done, gw = _to_sink_w(w, [x], [gy], False)
                    if not done:
                        gw = ConvWgrad.apply(x, gy, tuple(w.shape))
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    done, gb = _to_sink(ctx.bias, lambda out, acc: channel_sum_raw(gy, out=out, accum=acc))
                    if not done:
                        gb = ChannelSum.apply(gy)
        return gx, gw, gb


class ConvDgrad(Function):
    """gx = conv_transpose(gy, w): the data gradient as a first-class differentiable op."""

    @staticmethod
    def forward(ctx, gy, w):
        ctx.save_for_backward(gy, w)
        return conv_dgrad_raw(gy, w)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        d_gy = d_w = None
        if ctx.needs_input_grad[0]:
            d_gy = Conv.apply(ggx, w, None)
        if ctx.needs_input_grad[1] and _param_grads_enabled:
            done, d_w = _to_sink_w(w, [ggx], [gy], False)
            if not done:
                d_w = ConvWgrad.apply(ggx, gy, tuple(w.shape))
        return d_gy, d_w


class ConvWgrad(Function):
    """gw = sum_m gy (x) x_shifted: the weight gradient as a differentiable op."""

    @staticmethod
    def forward(ctx, x, gy, wshape):
        ctx.save_for_backward(x, gy)
        ctx.wshape = wshape
        return conv_wgrad_raw(x, gy, wshape)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        d_x = d_gy = None
        if ctx.needs_input_grad[0]:
            d_x = ConvDgrad.apply(gy, ggw)
        if ctx.needs_input_grad[1]:
            d_gy = Conv.apply(x, ggw, None)
        return d_x, d_gy, None


class ReluConv(Function):
    """y = conv(relu(x), w) + b with the ReLU applied while gathering (no activated copy is written):
    the `ReLU -> conv` pairs of DownBlock / res_block (layers.py:230-233, resnet3d.py:14-15).
    Backward: gx = dgrad(gy, w) * [x > 0];  gw = wgrad(relu(x), gy) with the same fused gather."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias, ctx.bias = b is not None, b
        return conv_fwd_raw(x, w, b, relu_in=True)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ConvDgradMaskG.apply(w, 1, gy, x)[0]          # dgrad(gy, w) * [x > 0] in one launch
        if _param_grads_enabled:
            both = False
            if ctx.needs_input_grad[1] and ctx.has_bias and ctx.needs_input_grad[2]:
                both, gw, gb = _to_sink_wb(w, ctx.bias, [x], [gy], True)
            if not both:
                if ctx.needs_input_grad[1]:
                    done, gw = _to_sink_w(w, [x], [gy], True)
                    if not done:
                        gw = ReluConvWgrad.apply(x, gy, tuple(w.shape))
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    done, gb = _to_sink(ctx.bias, lambda out, acc: channel_sum_raw(gy, out=out, accum=acc))
                    if not done:
                        gb = ChannelSum.apply(gy)
        return gx, gw, gb


class ReluConvWgrad(Function):
    """gw = wgrad(relu(x), gy). Its adjoints: d_x = dgrad(gy, ggw) * [x > 0], d_gy = conv(relu(x), ggw)."""

    @staticmethod
    def forward(ctx, x, gy, wshape):
        ctx.save_for_backward(x, gy)
        return conv_wgrad_raw(x, gy, wshape, relu_in=True)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        d_x = d_gy = None
        if ctx.needs_input_grad[0]:
            d_x = ReluMask.apply(ConvDgrad.apply(gy, ggw), x)
        if ctx.needs_input_grad[1]:
            d_gy = ReluConv.apply(x, ggw, None)
        return d_x, d_gy, None


class ChannelSum(Function):
    @staticmethod
    def forward(ctx, t):
        ctx.shape = tuple(t.shape)
        return channel_sum_raw(t)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        shp = ctx.shape
        rows = shp[0] * shp[1]
        S = 1
        for s in shp[2:]:
            S *= s
        gg = g.unsqueeze(0).expand(shp[0], shp[1]).contiguous()
        out = torch.empty(shp, device=g.device, dtype=torch.float32)
        check(lib().t2v_rowbcast(_p(gg), _p(out), rows, S, _stream()), 't2v_rowbcast')
        return out


def conv(x, w, b=None):
    """Differentiable stride-1 same-padded convolution for [N,C,D,H,W], [N,C,H,W] or [N,C] inputs."""
    dim = x.dim()
    y = Conv.apply(_as5(x), _as5(w), b)
    if dim == 4:
        return y.squeeze(2)
    if dim == 2:
        return y.view(y.size(0), y.size(1))
    return y


def relu_conv(x, w, b=None):
    """conv(relu(x), w) + b with the ReLU fused into the gather."""
    dim = x.dim()
    y = ReluConv.apply(_as5(x), _as5(w), b)
    return y.squeeze(2) if dim == 4 else y


def linear(x, w, b=None):
    """F.linear for [B,Cin] inputs == 1x1x1 convolution on a single voxel."""
    return conv(x, w, b)


# ------------------------------------------------------------------------------------------------
# element-wise / pooling Functions (closed under differentiation)
# ------------------------------------------------------------------------------------------------

def _ew(fn, name, *tensors):
    n = tensors[0].numel()
    ts = [_c(t) for t in tensors]
    out = torch.empty_like(ts[0])
    check(fn(*[_p(t) for t in ts], _p(out), n, _stream()), name)
    return out


class Relu(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _ew(lib().t2v_relu, 't2v_relu', x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return ReluMask.apply(g, x)


class ReluMask(Function):
    """g * [x > 0] — linear in g; its adjoint is itself."""

    @staticmethod
    def forward(ctx, g, x):
        ctx.save_for_backward(x)
        return _ew(lib().t2v_relu_mask, 't2v_relu_mask', g, x)

    @staticmethod
    def backward(ctx, gg):
        x, = ctx.saved_tensors
        return ReluMask.apply(gg, x), None


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return _ew(lib().t2v_add, 't2v_add', a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def relu(x):
    return Relu.apply(x)


def add(a, b):
    return Add.apply(a, b)


def _pool_args(k, s, p):
    A = C.c_int32 * 3
    return A(*k), A(*s), A(*p)


def _pool_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class AvgPool3d(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = _c(x)
        N, Cc, D, H, W = x.shape
        Do, Ho, Wo = _pool_out(D, k[0], s[0], p[0]), _pool_out(H, k[1], s[1], p[1]), _pool_out(W, k[2], s[2], p[2])
        ctx.cfg = (k, s, p, (D, H, W))
        y = torch.empty((N, Cc, Do, Ho, Wo), device=x.device, dtype=torch.float32)
        ka, sa, pa = _pool_args(k, s, p)
        check(lib().t2v_avgpool3d(_p(x), None, _p(y), N * Cc, D, H, W, Do, Ho, Wo, ka, sa, pa, _stream()), 't2v_avgpool3d')
        return y

    @staticmethod
    def backward(ctx, g):
        k, s, p, in_sp = ctx.cfg
        return AvgPool3dBwd.apply(g, k, s, p, in_sp), None, None, None


class AddAvgPool3d(Function):
    """pool(a + b) in one kernel (DownBlock: DownSample(skip) + DownSample(main) == DownSample(skip + main))."""

    @staticmethod
    def forward(ctx, a, b, k, s, p):
        a, b = _c(a), _c(b)
        N, Cc, D, H, W = a.shape
        Do, Ho, Wo = _pool_out(D, k[0], s[0], p[0]), _pool_out(H, k[1], s[1], p[1]), _pool_out(W, k[2], s[2], p[2])
        ctx.cfg = (k, s, p, (D, H, W))
        y = torch.empty((N, Cc, Do, Ho, Wo), device=a.device, dtype=torch.float32)
        ka, sa, pa = _pool_args(k, s, p)
        check(lib().t2v_avgpool3d(_p(a), _p(b), _p(y), N * Cc, D, H, W, Do, Ho, Wo, ka, sa, pa, _stream()), 't2v_avgpool3d')
        return y

    @staticmethod
    def backward(ctx, g):
        k, s, p, in_sp = ctx.cfg
        gx = AvgPool3dBwd.apply(g, k, s, p, in_sp)
        return gx, gx, None, None, None


def add_avg_pool3d(a, b, k, s, p=(0, 0, 0)):
    return AddAvgPool3d.apply(a, b, tuple(k), tuple(s), tuple(p))


class AvgPool3dBwd(Function):
    @staticmethod
    def forward(ctx, g, k, s, p, in_sp):
        g = _c(g)
        N, Cc, Do, Ho, Wo = g.shape
        D, H, W = in_sp
        ctx.cfg = (k, s, p)
        gx = torch.empty((N, Cc, D, H, W), device=g.device, dtype=torch.float32)
        ka, sa, pa = _pool_args(k, s, p)
        check(lib().t2v_avgpool3d_bwd(_p(g), _p(gx), N * Cc, D, H, W, Do, Ho, Wo, ka, sa, pa, _stream()),
              't2v_avgpool3d_bwd')
        return gx

    @staticmethod
    def backward(ctx, gg):
        k, s, p = ctx.cfg
        return AvgPool3d.apply(gg, k, s, p), None, None, None, None


def _pool_jobs(ins, in2s, outs, cfgs, in_sps, adds=None):
    """t2v_pool_job array: cfgs[i] = (k, s, p); in_sps[i] = (D, H, W) of the un-pooled tensor; NC from outs[i]."""
    from ._lib import PoolJob
    arr = (PoolJob * len(ins))()
    for i, a in enumerate(arr):
        k, s_, p = cfgs[i]
        D, H, W = in_sps[i]
        a.x, a.y = ins[i].data_ptr(), outs[i].data_ptr()
        a.x2 = in2s[i].data_ptr() if in2s is not None else None
        a.add = adds[i].data_ptr() if adds is not None else None
        a.NC = outs[i].shape[0] * outs[i].shape[1]
        a.D, a.H, a.W = D, H, W
        a.Do, a.Ho, a.Wo = _pool_out(D, k[0], s_[0], p[0]), _pool_out(H, k[1], s_[1], p[1]), _pool_out(W, k[2], s_[2], p[2])
        for j in range(3):
            a.k[j], a.s[j], a.p[j] = k[j], s_[j], p[j]
    return arr


class AvgPool3dG(Function):
    """ys[i] = pool_i(xs[i] (+ x2s[i])) (+ adds[i]) for several tensors (the pyramid levels of a DownBlock) in ONE launch.
    args: cfgs (tuple of (k, s, p)), has_second, has_add, then the xs, the x2s, the adds. Members whose output receives
    no gradient cost nothing in the backward."""

    @staticmethod
    def forward(ctx, cfgs, has2, has_add, *ts):
        n = len(cfgs)
        xs = [_c(t) for t in ts[:n]]
        x2s = [_c(t) for t in ts[n:2 * n]] if has2 else None
        adds = [_c(t) for t in ts[(2 * n if has2 else n):]] if has_add else None
        in_sps = [tuple(x.shape[2:]) for x in xs]
        ys = []
        for x, (k, s_, p) in zip(xs, cfgs):
            ys.append(torch.empty((x.shape[0], x.shape[1], _pool_out(x.shape[2], k[0], s_[0], p[0]),
                                   _pool_out(x.shape[3], k[1], s_[1], p[1]), _pool_out(x.shape[4], k[2], s_[2], p[2])),
                                  device=x.device, dtype=torch.float32))
        if adds is not None and any(a.shape != y.shape for a, y in zip(adds, ys)):
            raise ValueError('addend / pooled shape mismatch')
        check(lib().t2v_avgpool3d_multi(_pool_jobs(xs, x2s, ys, cfgs, in_sps, adds), n, _stream()), 't2v_avgpool3d_multi')
        ctx.cfg = (cfgs, has2, has_add, in_sps)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        cfgs, has2, has_add, in_sps = ctx.cfg
        n = len(cfgs)
        live = [i for i, g in enumerate(gs) if g is not None]
        gxs = [None] * n
        g_add = gs
        if has_add and torch.is_grad_enabled():      # recorded backward (gradient penalty): dL/dy feeds the pooling adjoint AND the
            gs, g_add = _fork_some(gs)               # addend; a grouped fork sums their second-order gradients in one launch
        if live:
            res = AvgPool3dBwdG.apply(tuple(cfgs[i] for i in live), tuple(in_sps[i] for i in live), *[gs[i] for i in live])
            for i, r in zip(live, res):
                gxs[i] = r
        gx2s = gxs
        if has2 and torch.is_grad_enabled():         # likewise the one gradient that goes to both pooled inputs
            gxs, gx2s = _fork_some(gxs)
        return (None, None, None) + tuple(gxs) + (tuple(gx2s) if has2 else ()) + (tuple(g_add) if has_add else ())


class AvgPool3dBwdG(Function):
    @staticmethod
    def forward(ctx, cfgs, in_sps, *gs):
        gs = [_c(g) for g in gs]
        gxs = [torch.empty((g.shape[0], g.shape[1]) + tuple(sp), device=g.device, dtype=torch.float32) for g, sp in zip(gs, in_sps)]
        check(lib().t2v_avgpool3d_bwd_multi(_pool_jobs(gs, None, gxs, cfgs, in_sps), len(gs), _stream()), 't2v_avgpool3d_bwd_multi')
        ctx.cfg = cfgs
        ctx.set_materialize_grads(False)
        return tuple(gxs)

    @staticmethod
    def backward(ctx, *ggs):
        cfgs = ctx.cfg
        live = [i for i, g in enumerate(ggs) if g is not None]
        out = [None] * len(cfgs)
        if live:
            res = AvgPool3dG.apply(tuple(cfgs[i] for i in live), False, False, *[ggs[i] for i in live])
            for i, r in zip(live, res):
                out[i] = r
        return (None, None) + tuple(out)


class ForkPoolG(Function):
    """(xs', pool(xs)) for n tensors that feed a second consumer besides the pooling (a DownBlock's input: main path + pooled skip
    path): outputs the n aliases, then the n pooled tensors (one launch). In the plain backward the adjoint of the pooling ADDS the
    alias' gradient in its own launch (`t2v_pool_job.add`) — no separate grouped add, one full-resolution tensor less through HBM;
    the recorded backward (gradient penalty) composes the differentiable pieces."""

    @staticmethod
    def forward(ctx, cfgs, *xs):
        xs = [_c(x) for x in xs]
        in_sps = [tuple(x.shape[2:]) for x in xs]
        ys = [torch.empty((x.shape[0], x.shape[1], _pool_out(x.shape[2], k[0], s_[0], p[0]), _pool_out(x.shape[3], k[1], s_[1], p[1]),
                           _pool_out(x.shape[4], k[2], s_[2], p[2])), device=x.device, dtype=torch.float32) for x, (k, s_, p) in zip(xs, cfgs)]
        check(lib().t2v_avgpool3d_multi(_pool_jobs(xs, None, ys, cfgs, in_sps), len(xs), _stream()), 't2v_avgpool3d_multi')
        ctx.cfg = (cfgs, in_sps)
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs) + tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        cfgs, in_sps = ctx.cfg
        n = len(cfgs)
        ga, gp = list(gs[:n]), list(gs[n:])
        out = list(ga)
        live = [i for i in range(n) if gp[i] is not None]
        if live and torch.is_grad_enabled():
            res = AvgPool3dBwdG.apply(tuple(cfgs[i] for i in live), tuple(in_sps[i] for i in live), *[gp[i] for i in live])
            both = [k for k, i in enumerate(live) if ga[i] is not None]
            for k, i in enumerate(live):
                if ga[i] is None:
                    out[i] = res[k]
            if both:
                for k, t in zip(both, AddG.apply(*([ga[live[k]] for k in both] + [res[k] for k in both]))):
                    out[live[k]] = t
        elif live:
            gpl = [_c(gp[i]) for i in live]
            adds = [(_c(ga[i]) if ga[i] is not None else None) for i in live]
            gxs = [torch.empty((g.shape[0], g.shape[1]) + tuple(in_sps[i]), device=g.device, dtype=torch.float32) for g, i in zip(gpl, live)]
            jobs = _pool_jobs(gpl, None, gxs, [cfgs[i] for i in live], [in_sps[i] for i in live])
            for a, ad in zip(jobs, adds):
                a.add = ad.data_ptr() if ad is not None else None
            check(lib().t2v_avgpool3d_bwd_multi(jobs, len(live), _stream()), 't2v_avgpool3d_bwd_multi')
            for i, t in zip(live, gxs):
                out[i] = t
        return (None,) + tuple(out)


def fork_pool_group(xs, cfgs):
    """(aliases of xs, [pool_i(xs[i])]) with the aliases' gradients summed INSIDE the pooling adjoint's launch. Falls back to a
    grouped fork + `avg_pool3d_group` for padded / overlapping windows and for tensors that need no gradient."""
    xs = list(xs)
    cfgs = tuple((tuple(k), tuple(s_), tuple(p)) for k, s_, p in cfgs)
    plain = any(any(p) or any(kk > ss for kk, ss in zip(k, s_)) for k, s_, p in cfgs) or len(xs) > 8
    if plain or not torch.is_grad_enabled() or not all(x.requires_grad for x in xs):
        xa, xb = fork_group(xs)
        return xa, avg_pool3d_group(xb, cfgs)
    res = ForkPoolG.apply(cfgs, *xs)
    return list(res[:len(xs)]), list(res[len(xs):])


def avg_pool3d_group(xs, cfgs, x2s=None, adds=None):
    """cfgs[i] = (k, s, p) per tensor; x2s: pool(xs[i] + x2s[i]); adds: ... + adds[i] (pooled shape). Falls back to
    per-tensor launches for padded / overlapping windows (their adjoint is the general gather kernel)."""
    cfgs = tuple((tuple(k), tuple(s_), tuple(p)) for k, s_, p in cfgs)
    if any(any(p) or any(kk > ss for kk, ss in zip(k, s_)) for k, s_, p in cfgs) or len(xs) > 8:
        if x2s is None:
            ys = [avg_pool3d(x, *c) for x, c in zip(xs, cfgs)]
        else:
            ys = [add_avg_pool3d(x, x2, *c) for x, x2, c in zip(xs, x2s, cfgs)]
        return ys if adds is None else [add(y, a) for y, a in zip(ys, adds)]
    ts = list(xs) + (list(x2s) if x2s is not None else []) + (list(adds) if adds is not None else [])
    return list(AvgPool3dG.apply(cfgs, x2s is not None, adds is not None, *ts))


def avg_pool3d(x, k, s, p=(0, 0, 0)):
    return AvgPool3d.apply(x, tuple(k), tuple(s), tuple(p))


class MaxPool2x2(Function):
    """2x2 max-pool over the trailing (H,W) plane of any tensor (F.max_pool2d [2,2] / max_pool3d [1,2,2])."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        H, W = x.shape[-2], x.shape[-1]
        planes = x.numel() // (H * W)
        y = torch.empty(tuple(x.shape[:-2]) + (H // 2, W // 2), device=x.device, dtype=torch.float32)
        idx = torch.empty(y.shape, device=x.device, dtype=torch.int32)
        check(lib().t2v_maxpool2x2(_p(x), _p(y), _p(idx), planes, H, W, _stream()), 't2v_maxpool2x2')
        ctx.save_for_backward(idx)
        ctx.in_shape = tuple(x.shape)
        ctx.mark_non_differentiable(idx)
        return y

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        return MaxScatter.apply(g, idx, ctx.in_shape)


class MaxScatter(Function):
    @staticmethod
    def forward(ctx, g, idx, in_shape):
        g = _c(g)
        H, W = in_shape[-2], in_shape[-1]
        planes = g.numel() // ((H // 2) * (W // 2))
        gx = torch.empty(in_shape, device=g.device, dtype=torch.float32)
        check(lib().t2v_maxpool2x2_scatter(_p(g), _p(idx), _p(gx), planes, H, W, _stream()), 't2v_maxpool2x2_scatter')
        ctx.save_for_backward(idx)
        ctx.in_shape = in_shape
        return gx

    @staticmethod
    def backward(ctx, gg):
        idx, = ctx.saved_tensors
        return MaxGather.apply(gg, idx, ctx.in_shape), None, None


class MaxGather(Function):
    @staticmethod
    def forward(ctx, x, idx, in_shape):
        x = _c(x)
        H, W = in_shape[-2], in_shape[-1]
        planes = x.numel() // (H * W)
        y = torch.empty(tuple(in_shape[:-2]) + (H // 2, W // 2), device=x.device, dtype=torch.float32)
        check(lib().t2v_maxpool2x2_gather(_p(x), _p(idx), _p(y), planes, H, W, _stream()), 't2v_maxpool2x2_gather')
        ctx.save_for_backward(idx)
        ctx.in_shape = in_shape
        return y

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        return MaxScatter.apply(g, idx, ctx.in_shape), None, None


def max_pool2x2(x):
    return MaxPool2x2.apply(x)


class RowSum(Function):
    """[..., S] -> [...] : sum over the trailing flattened voxels (torch.sum(x,[2,3,4]))."""

    @staticmethod
    def forward(ctx, x, lead):
        x = _c(x)
        shp = tuple(x.shape)
        rows = 1
        for s in shp[:lead]:
            rows *= s
        S = x.numel() // rows
        ctx.shape = shp
        ctx.lead = lead
        y = torch.empty(shp[:lead], device=x.device, dtype=torch.float32)
        check(lib().t2v_rowsum(_p(x), _p(y), rows, S, _stream()), 't2v_rowsum')
        return y

    @staticmethod
    def backward(ctx, g):
        return RowBcast.apply(g, ctx.shape, ctx.lead), None


class RowBcast(Function):
    @staticmethod
    def forward(ctx, g, shape, lead):
        g = _c(g)
        rows = g.numel()
        S = 1
        for s in shape[lead:]:
            S *= s
        ctx.lead = lead
        out = torch.empty(shape, device=g.device, dtype=torch.float32)
        check(lib().t2v_rowbcast(_p(g), _p(out), rows, S, _stream()), 't2v_rowbcast')
        return out

    @staticmethod
    def backward(ctx, gg):
        return RowSum.apply(gg, ctx.lead), None, None


def sum_spatial(x):
    """[b,C,T,H,W] -> [b,C]."""
    return RowSum.apply(x, 2)


# ------------------------------------------------------------------------------------------------
# non-local block pieces
# ------------------------------------------------------------------------------------------------

def _bmm_raw(A, B, M, N, K, ta, tb):
    A, B = _c(A), _c(B)
    batch = A.shape[0]
    out = torch.empty((batch, M, N), device=A.device, dtype=torch.float32)
    check(lib().t2v_bmm(_p(A), _p(B), _p(out), batch, M, N, K, int(ta), int(tb), 0, _stream()), 't2v_bmm')
    return out


class Bmm(Function):
    """C[b] = op(A[b]) @ op(B[b]); A stored [b,M,K] (ta=0) or [b,K,M] (ta=1); B [b,K,N] / [b,N,K]."""

    @staticmethod
    def forward(ctx, A, B, ta, tb):
        M = A.shape[2] if ta else A.shape[1]
        K = A.shape[1] if ta else A.shape[2]
        N = B.shape[1] if tb else B.shape[2]
        ctx.save_for_backward(A, B)
        ctx.cfg = (ta, tb)
        return _bmm_raw(A, B, M, N, K, ta, tb)

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        ta, tb = ctx.cfg
        dA = dB = None
        if ctx.needs_input_grad[0]:
            if not ta:
                dA = Bmm.apply(G, B, False, not tb)          # [M,N] x B'^T
            else:
                dA = Bmm.apply(B, G, tb, True)               # B' x G^T  -> [K,M]
        if ctx.needs_input_grad[1]:
            if not tb:
                dB = Bmm.apply(A, G, not ta, False)          # A'^T x G -> [K,N]
            else:
                dB = Bmm.apply(G, A, True, ta)               # G^T x A' -> [N,K]
        return dA, dB, None, None


def bmm(A, B, ta=False, tb=False):
    return Bmm.apply(A, B, ta, tb)


class Softmax(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n = x.shape[-1]
        y = torch.empty_like(x)
        check(lib().t2v_softmax(_p(x), _p(y), x.numel() // n, n, _stream()), 't2v_softmax')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return SoftmaxBwd.apply(y, g)


class SoftmaxBwd(Function):
    @staticmethod
    def forward(ctx, y, gy):
        y, gy = _c(y), _c(gy)
        n = y.shape[-1]
        gx = torch.empty_like(y)
        check(lib().t2v_softmax_bwd(_p(y), _p(gy), _p(gx), y.numel() // n, n, _stream()), 't2v_softmax_bwd')
        ctx.save_for_backward(y, gy)
        return gx

    @staticmethod
    def backward(ctx, gg):
        y, gy = ctx.saved_tensors
        gg = _c(gg)
        n = y.shape[-1]
        d_y = d_gy = None
        if ctx.needs_input_grad[0]:
            d_y = torch.empty_like(y)
            check(lib().t2v_softmax_bwd_bwd_y(_p(y), _p(gy), _p(gg), _p(d_y), y.numel() // n, n, _stream()),
                  't2v_softmax_bwd_bwd_y')
        if ctx.needs_input_grad[1]:
            d_gy = SoftmaxBwd.apply(y, gg)
        return d_y, d_gy


def softmax_lastdim(x):
    return Softmax.apply(x)


class NonlocalAttend(Function):
    """o = g . softmax(theta^T phi)^T fused (`t2v_nonlocal_fwd/_bwd`): beta [b, N, Nk] never reaches HBM. First-order only
    (the generator's 2-D block, layers.py:23-36; the discriminator's 3-D block sits inside the gradient penalty's recorded
    graph and stays on the closed bmm / softmax set). theta [b,C8,N], phi [b,C8,Nk], g [b,C2,Nk] -> o [b,C2,N]."""

    @staticmethod
    def forward(ctx, theta, phi, g):
        theta, phi, g = _c(theta), _c(phi), _c(g)
        b, C8, N = theta.shape
        C2, Nk = g.shape[1], g.shape[2]
        o = torch.empty((b, C2, N), device=theta.device, dtype=torch.float32)
        lse = torch.empty((b, N), device=theta.device, dtype=torch.float32)
        check(lib().t2v_nonlocal_fwd(_p(theta), _p(phi), _p(g), _p(o), _p(lse), b, C8, C2, N, Nk, _stream()), 't2v_nonlocal_fwd')
        ctx.save_for_backward(theta, phi, g, o, lse)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        theta, phi, g, o, lse = ctx.saved_tensors
        go = _c(go)
        b, C8, N = theta.shape
        C2, Nk = g.shape[1], g.shape[2]
        dtheta, dphi, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(g)
        ws = torch.empty((b, N), device=theta.device, dtype=torch.float32)
        check(lib().t2v_nonlocal_bwd(_p(theta), _p(phi), _p(g), _p(o), _p(lse), _p(go), _p(dtheta), _p(dphi), _p(dg), _p(ws),
                                     b, C8, C2, N, Nk, _stream()), 't2v_nonlocal_bwd')
        return dtheta, dphi, dg


def nonlocal_attend_ok(c8, c2):
    return bool(lib().t2v_nonlocal_ok(int(c8), int(c2)))


def nonlocal_attend(theta, phi, g):
    return NonlocalAttend.apply(theta, phi, g)


class Dot(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = torch.empty((256,), device=a.device, dtype=torch.float32)
        check(lib().t2v_dot(_p(a), _p(b), _p(out), _p(ws), a.numel(), 0, _stream()), 't2v_dot')
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ScaleDev.apply(g, b), ScaleDev.apply(g, a)


class ScaleDev(Function):
    """y = s * a with s a 0-d device tensor (the non-local gain `gamma`)."""

    @staticmethod
    def forward(ctx, s, a):
        a = _c(a)
        ctx.save_for_backward(s, a)
        y = torch.empty_like(a)
        check(lib().t2v_scale_dev(_p(s), 1.0, _p(a), _p(y), a.numel(), _stream()), 't2v_scale_dev')
        return y

    @staticmethod
    def backward(ctx, g):
        s, a = ctx.saved_tensors
        gs = ga = None
        if ctx.needs_input_grad[0]:
            gs = Dot.apply(g, a)
        if ctx.needs_input_grad[1]:
            ga = ScaleDev.apply(s, g)
        return gs, ga


def scale_add(gamma, o, x):
    """gamma * o + x  (layers.py:36,68)."""
    return Add.apply(ScaleDev.apply(gamma, o), x)


# ------------------------------------------------------------------------------------------------
# the non-local block's small ops over several tensors at once (`t2v_multi`) and the grouped fork / add / cat-lerp glue live in
# functional_multi.py; every name is re-exported here (call sites use `functional.<name>`)
# ------------------------------------------------------------------------------------------------
from .functional_multi import *          # noqa: E402,F401,F403
from .functional_multi import _mj, _live, _fork_some, _bmm_dims          # noqa: E402,F401


# ------------------------------------------------------------------------------------------------
# generator-side ops (first-order only)
# ------------------------------------------------------------------------------------------------

class Tanh(Function):
    @staticmethod
    def forward(ctx, x):
        y = _ew(lib().t2v_tanh, 't2v_tanh', x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y, = ctx.saved_tensors
        return _ew(lib().t2v_tanh_bwd, 't2v_tanh_bwd', g, y)


def tanh(x):
    return Tanh.apply(x)


class Upsample2x(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        H, W = x.shape[-2], x.shape[-1]
        planes = x.numel() // (H * W)
        y = torch.empty(tuple(x.shape[:-2]) + (2 * H, 2 * W), device=x.device, dtype=torch.float32)
        check(lib().t2v_upsample2x(_p(x), _p(y), planes, H, W, _stream()), 't2v_upsample2x')
        ctx.in_shape = tuple(x.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g)
        shp = ctx.in_shape
        H, W = shp[-2], shp[-1]
        gx = torch.empty(shp, device=g.device, dtype=torch.float32)
        check(lib().t2v_upsample2x_bwd(_p(g), _p(gx), gx.numel() // (H * W), H, W, _stream()), 't2v_upsample2x_bwd')
        return gx


def upsample2x(x):
    return Upsample2x.apply(x)


class UpsampleAdd(Function):
    """upsample2x(s) + h in one launch (UpBlock's residual add with the identity path's up-sampling folded in)."""

    @staticmethod
    def forward(ctx, s, h):
        s, h = _c(s), _c(h)
        H, W = s.shape[-2], s.shape[-1]
        if tuple(h.shape) != tuple(s.shape[:-2]) + (2 * H, 2 * W):
            raise ValueError('upsample_add: %s vs %s' % (tuple(s.shape), tuple(h.shape)))
        y = torch.empty_like(h)
        check(lib().t2v_upsample2x_add(_p(s), _p(h), _p(y), s.numel() // (H * W), H, W, _stream()), 't2v_upsample2x_add')
        ctx.in_shape = tuple(s.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        gs = None
        if ctx.needs_input_grad[0]:
            g = _c(g)
            shp = ctx.in_shape
            H, W = shp[-2], shp[-1]
            gs = torch.empty(shp, device=g.device, dtype=torch.float32)
            check(lib().t2v_upsample2x_bwd(_p(g), _p(gs), gs.numel() // (H * W), H, W, _stream()), 't2v_upsample2x_bwd')
        return gs, (g if ctx.needs_input_grad[1] else None)


def upsample_add(s, h):
    return UpsampleAdd.apply(s, h)


class BatchNormAct(Function):
    """BatchNorm2d (+ optional fused ReLU, + optional fused nearest x2 up-sampling of the result: `up`, 4-D inputs).
    Training: batch statistics, running stats updated in place (momentum 0.1, unbiased variance); eval: running
    statistics (no grad support needed)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, momentum, eps, relu, counter=None, up=False):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        if up and x.dim() != 4:
            raise ValueError('the fused up-sampling takes [N,C,H,W] inputs')
        ctx.up = bool(up)
        if training:
            stats = torch.empty((2 * Cc,), device=x.device, dtype=torch.float32)
            ws = torch.empty((int(lib().t2v_bn_ws_floats(N, Cc, S)),), device=x.device, dtype=torch.float32)
            if up:
                H, W = x.shape[2], x.shape[3]
                y = torch.empty((N, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
                check(lib().t2v_bn_train_fwd_up(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(rmean), _p(rvar), _p(ws), N, Cc, H, W,
                                                momentum, eps, int(relu), _p(counter), _stream()), 't2v_bn_train_fwd_up')
            else:
                y = torch.empty_like(x)
                check(lib().t2v_bn_train_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(rmean), _p(rvar), _p(ws), N, Cc, S,
                                             momentum, eps, int(relu), _p(counter), _stream()), 't2v_bn_train_fwd')
            ctx.save_for_backward(x, y, stats, gamma)
            ctx.relu = relu
        else:
            y = torch.empty_like(x)
            check(lib().t2v_bn_eval(_p(x), _p(rmean), _p(rvar), _p(gamma), _p(beta), _p(y), N, Cc, S, eps, int(relu), _stream()),
                  't2v_bn_eval')
            if up:                  # (sampling path: two launches)
                H, W = x.shape[2], x.shape[3]
                y2 = torch.empty((N, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
                check(lib().t2v_upsample2x(_p(y), _p(y2), N * Cc, H, W, _stream()), 't2v_upsample2x')
                y = y2
        ctx.training = training
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if not ctx.training:
            raise RuntimeError('eval-mode BatchNorm backward is not on the hot path')
        x, y, stats, gamma = ctx.saved_tensors
        g = _c(g)
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        gx = torch.empty_like(x)
        gg = torch.empty_like(gamma)
        gb = torch.empty_like(gamma)
        ws = torch.empty((int(lib().t2v_bn_ws_floats(N, Cc, S)),), device=x.device, dtype=torch.float32)
        if ctx.up:
            check(lib().t2v_bn_train_bwd_up(_p(g), _p(x), _p(y), _p(stats), _p(gamma), _p(gx), _p(gg), _p(gb), _p(ws), N, Cc, x.shape[2],
                                            x.shape[3], int(ctx.relu), _stream()), 't2v_bn_train_bwd_up')
        else:
            check(lib().t2v_bn_train_bwd(_p(g), _p(x), _p(y), _p(stats), _p(gamma), _p(gx), _p(gg), _p(gb), _p(ws), N, Cc, S,
                                         int(ctx.relu), _stream()), 't2v_bn_train_bwd')
        return gx, gg, gb, None, None, None, None, None, None, None, None


class BatchNormActFork(Function):
    """Training-mode `BatchNormAct` of a tensor that ALSO feeds a second consumer (an UpBlock's input: BatchNorm + skip path,
    layers.py:152-195; the map in front of a RenderBlock: BatchNorm + the next level): returns (y, x') with x' an alias of x for
    the other consumer. The adjoint receives both gradients at once and the BatchNorm backward pass adds the alias's gradient
    while it writes dL/dx (`t2v_bn_train_bwd_add`): no add launch of the autograd engine, no add launch at all."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, momentum, eps, relu, counter, up):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        if x.dim() != 4:
            raise ValueError('BatchNorm2d takes [N,C,H,W] inputs')
        ctx.up, ctx.relu = bool(up), relu
        stats = torch.empty((2 * Cc,), device=x.device, dtype=torch.float32)
        ws = torch.empty((int(lib().t2v_bn_ws_floats(N, Cc, S)),), device=x.device, dtype=torch.float32)
        H, W = x.shape[2], x.shape[3]
        if up:
            y = torch.empty((N, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
            check(lib().t2v_bn_train_fwd_up(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(rmean), _p(rvar), _p(ws), N, Cc, H, W,
                                            momentum, eps, int(relu), _p(counter), _stream()), 't2v_bn_train_fwd_up')
        else:
            y = torch.empty_like(x)
            check(lib().t2v_bn_train_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(rmean), _p(rvar), _p(ws), N, Cc, S,
                                         momentum, eps, int(relu), _p(counter), _stream()), 't2v_bn_train_fwd')
        ctx.save_for_backward(x, y, stats, gamma)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g2):
        x, y, stats, gamma = ctx.saved_tensors
        if g is None:                                   # (only the alias was used)
            return (g2,) + (None,) * 9
        g = _c(g)
        g2 = _c(g2) if g2 is not None else None
        N, Cc, H, W = x.shape
        gx = torch.empty_like(x)
        gg = torch.empty_like(gamma)
        gb = torch.empty_like(gamma)
        ws = torch.empty((int(lib().t2v_bn_ws_floats(N, Cc, H * W)),), device=x.device, dtype=torch.float32)
        check(lib().t2v_bn_train_bwd_add(_p(g), _p(x), _p(y), _p(stats), _p(gamma), _p(g2), _p(gx), _p(gg), _p(gb), _p(ws), N, Cc, H, W,
                                         int(ctx.up), int(ctx.relu), _stream()), 't2v_bn_train_bwd_add')
        return gx, gg, gb, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, rmean, rvar, training, momentum=0.1, eps=1e-5, relu=False, counter=None, up=False, fork=False):
    """`counter`: the module's int64 `num_batches_tracked` buffer, incremented by the same launch in training mode.
    `up`: the result goes through a nearest x2 up-sampling in the same launches (UpBlock's BN-ReLU-Up head).
    `fork`: returns (y, x') — x' an alias of the input for its OTHER consumer, whose gradient the BatchNorm adjoint sums in its
    own pass (`BatchNormActFork`; eval mode / 3-D inputs: x' is x itself)."""
    if counter is not None and (counter.dtype != torch.int64 or not counter.is_cuda):
        raise TypeError('num_batches_tracked must be an int64 device tensor')
    if fork:
        if training and x.dim() == 4 and x.requires_grad and torch.is_grad_enabled():
            return BatchNormActFork.apply(x, gamma, beta, rmean, rvar, momentum, eps, relu, counter, up)
        return BatchNormAct.apply(x, gamma, beta, rmean, rvar, training, momentum, eps, relu, counter, up), x
    return BatchNormAct.apply(x, gamma, beta, rmean, rvar, training, momentum, eps, relu, counter, up)


def _skinny_ok(M, K, N):
    return M <= 64 and int(lib().t2v_skinny_gemm_splits(M, K, N)) > 0


def _lstm_fused_ok(B, K, Cc):
    return os.environ.get('T2V_NO_LSTM_FUSED', '0') != '1' and bool(lib().t2v_lstm_step_fused_ok(B, K, Cc))


def _skinny(x, wp, M, K, N):
    """slab[S][M][N] of partial products x[M][K] . wp[K][N] (`t2v_skinny_gemm_slab`); the consumer sums the S slices."""
    S = int(lib().t2v_skinny_gemm_splits(M, K, N))
    slab = torch.empty((S, M, N), device=x.device, dtype=torch.float32)
    check(lib().t2v_skinny_gemm_slab(_p(x), _p(wp), _p(slab), M, K, N, _stream()), 't2v_skinny_gemm_slab')
    return slab, S


class ConvLSTMFn(Function):
    """16-step single-cell ConvLSTM (conv_lstm.py:75-97). x is the input at step 0 and zero afterwards,
    so the Wx convolutions run once; from step 1 on their contribution is the bias. The four gate weights
    are packed side by side, so each step is ONE convolution with 4C output channels (and its data
    gradient ONE convolution with 4C input channels); the weight gradients of all gates and all time
    steps come from one launch. Back-propagation through time is done here explicitly.
    args: x, steps, then Wx{i,f,c,o}, bx{i,f,c,o}, Wh{i,f,c,o}  (4-D conv weights [C,C,3,3]).
    Returns the stacked hidden states [steps, B, C, h, w]."""

    @staticmethod
    def forward(ctx, x, steps, *params):
        wx, bx, wh = params[0:4], params[4:8], params[8:12]
        x = _c(x)
        B, Cc, h, w = x.shape
        CS = Cc * h * w
        k = (1,) + tuple(wx[0].shape[2:])
        x5 = x.unsqueeze(2)
        dev = x.device
        g0 = conv_geom(B, Cc, 1, h, w, 4 * Cc, k[0], k[1], k[2])
        ts = _tapset(g0.T, g0.mask)
        wx5 = [t.unsqueeze(2) for t in wx]
        wh5 = [t.unsqueeze(2) for t in wh]
        wpx = packed_fused(wx5, ts, 0)
        wph = packed_fused(wh5, ts, 0)
        bias4 = torch.empty((4 * Cc,), device=dev, dtype=torch.float32)
        check(lib().t2v_concat4(_p(_c(bx[0])), _p(_c(bx[1])), _p(_c(bx[2])), _p(_c(bx[3])), _p(bias4), Cc, _stream()), 't2v_concat4')
        hs = torch.empty((steps, B, Cc, 1, h, w), device=dev, dtype=torch.float32)
        cs = torch.empty((steps + 1, B, Cc, h, w), device=dev, dtype=torch.float32)
        acts = torch.empty((steps, B, 4 * Cc, h, w), device=dev, dtype=torch.float32)
        pre = torch.empty((B, 4 * Cc, 1, h, w), device=dev, dtype=torch.float32)
        check(lib().t2v_fill(_p(cs[0]), 0.0, B * CS, _stream()), 't2v_fill')
        if h == 1 and w == 1 and _lstm_fused_ok(B, Cc, Cc):
            # 1x1 maps (the TGANv2 generator): GEMM + gate math in ONE launch per step, on the unit-major weight copies
            wr = torch.empty((2, Cc * 4 * Cc), device=dev, dtype=torch.float32)
            check(lib().t2v_lstm_pack_major(_p(wpx), _p(wr[0]), Cc, Cc, _stream()), 't2v_lstm_pack_major')
            if steps > 1:
                check(lib().t2v_lstm_pack_major(_p(wph), _p(wr[1]), Cc, Cc, _stream()), 't2v_lstm_pack_major')
            for t in range(steps):
                check(lib().t2v_lstm_step_fused(_p(x5 if t == 0 else hs[t - 1]), _p(wr[0 if t == 0 else 1]), _p(bias4), _p(cs[t]),
                                                _p(hs[t]), _p(cs[t + 1]), _p(acts[t]), B, Cc, Cc, _stream()), 't2v_lstm_step_fused')
            ctx.save_for_backward(x, hs, cs, acts, *params)
            ctx.steps = steps
            return hs.squeeze(3)
        if h == 1 and w == 1 and _skinny_ok(B, Cc, 4 * Cc):
            # 1x1 maps (the TGANv2 generator): a step = one 32-row GEMM; wave-per-strip kernel + slab-summing gate kernel
            for t in range(steps):
                slab, S = _skinny(x5 if t == 0 else hs[t - 1], wpx if t == 0 else wph, B, Cc, 4 * Cc)
                check(lib().t2v_lstm_gates_slab(_p(slab), S, _p(bias4), _p(cs[t]), _p(hs[t]), _p(cs[t + 1]), _p(acts[t]), B, Cc,
                                                _stream()), 't2v_lstm_gates_slab')
            ctx.save_for_backward(x, hs, cs, acts, *params)
            ctx.steps = steps
            return hs.squeeze(3)
        for t in range(steps):
            if t == 0:
                conv_packed_raw(x5, wpx, Cc, 4 * Cc, k, bias4, out=pre)           # h0 = 0: the Wh term vanishes
            else:
                conv_packed_raw(hs[t - 1], wph, Cc, 4 * Cc, k, bias4, out=pre)    # x_t = 0: Wx(x_t) is its bias
            check(lib().t2v_lstm_gates(_p(pre), _p(cs[t]), _p(hs[t]), _p(cs[t + 1]), _p(acts[t]), B, CS, _stream()),
                  't2v_lstm_gates')
        ctx.save_for_backward(x, hs, cs, acts, *params)
        ctx.steps = steps
        return hs.squeeze(3)

    @staticmethod
    @once_differentiable
    def backward(ctx, ghs):
        saved = ctx.saved_tensors
        x, hs, cs, acts = saved[0:4]
        params = saved[4:]
        wx, bx, wh = params[0:4], params[4:8], params[8:12]
        steps = ctx.steps
        ghs = _c(ghs)
        B, Cc, h, w = x.shape
        CS = Cc * h * w
        dev = x.device
        k = (1,) + tuple(wx[0].shape[2:])
        x5 = x.unsqueeze(2)
        gt = conv_geom(B, 4 * Cc, 1, h, w, Cc, k[0], k[1], k[2])         # data-gradient geometry (4C -> C)
        ts = _tapset(gt.T, gt.mask)
        wx5 = [t.unsqueeze(2) for t in wx]
        wh5 = [t.unsqueeze(2) for t in wh]
        wph1 = packed_fused(wh5, ts, 1)
        gpre = torch.empty((steps, B, 4 * Cc, 1, h, w), device=dev, dtype=torch.float32)
        gh_next = None           # dL/dh_t arriving from step t+1
        gc = None
        fused = h == 1 and w == 1 and _lstm_fused_ok(B, Cc, Cc) and Cc % 64 == 0
        skinny = not fused and h == 1 and w == 1 and _skinny_ok(B, 4 * Cc, Cc)
        slab, S = None, 0
        if fused:                                                        # GEMM + gate adjoints in ONE launch per step
            gcs = torch.empty((2, B, Cc), device=dev, dtype=torch.float32)
            w1r = torch.empty((4 * Cc * Cc,), device=dev, dtype=torch.float32)
            if steps > 1:
                check(lib().t2v_lstm_pack_cols(_p(wph1), _p(w1r), 4 * Cc, Cc, _stream()), 't2v_lstm_pack_cols')
            for t in range(steps - 1, -1, -1):
                check(lib().t2v_lstm_step_bwd_fused(_p(ghs[t]), _p(gpre[t + 1]) if t + 1 < steps else None, _p(w1r),
                                                    _p(gc), _p(acts[t]), _p(cs[t]), _p(cs[t + 1]), _p(gpre[t]), _p(gcs[t & 1]), B, Cc,
                                                    _stream()), 't2v_lstm_step_bwd_fused')
                gc = gcs[t & 1]
        for t in range(steps - 1, -1, -1) if skinny else ():
            gcp = torch.empty((B, Cc, h, w), device=dev, dtype=torch.float32)
            check(lib().t2v_lstm_gates_bwd_slab(_p(ghs[t]), _p(slab), S, _p(gc), _p(acts[t]), _p(cs[t]), _p(cs[t + 1]), _p(gpre[t]),
                                                _p(gcp), B, Cc, _stream()), 't2v_lstm_gates_bwd_slab')
            gc = gcp
            if t > 0:
                slab, S = _skinny(gpre[t], wph1, B, 4 * Cc, Cc)
        for t in range(steps - 1, -1, -1) if not (skinny or fused) else ():
            if gh_next is None:
                gh = ghs[t]
            else:
                gh = torch.empty((B, Cc, h, w), device=dev, dtype=torch.float32)
                check(lib().t2v_add(_p(ghs[t]), _p(gh_next), _p(gh), B * CS, _stream()), 't2v_add')
            gcp = torch.empty((B, Cc, h, w), device=dev, dtype=torch.float32)
            check(lib().t2v_lstm_gates_bwd(_p(gh), _p(gc), _p(acts[t]), _p(cs[t]), _p(cs[t + 1]), _p(gpre[t]), _p(gcp),
                                           B, CS, _stream()), 't2v_lstm_gates_bwd')
            gc = gcp
            if t > 0:
                gh_next, _ = conv_packed_raw(gpre[t], wph1, 4 * Cc, Cc, k)
        gx = None
        if ctx.needs_input_grad[0]:
            wpx1 = packed_fused(wx5, ts, 1)
            gx, _ = conv_packed_raw(gpre[0], wpx1, 4 * Cc, Cc, k)
            gx = gx.squeeze(2)
        w4shape = (4 * Cc, Cc) + k
        gb4 = channel_sum_raw(gpre.view(steps * B, 4 * Cc, 1, h, w))      # every step contributes to the bias
        gbx = [gb4[g * Cc:(g + 1) * Cc] for g in range(4)]
        sink = _grad_sink
        if (sink is not None and sink.defer and not torch.is_grad_enabled() and steps > 1 and
                all(id(t) in sink.slots and not t.is_contiguous() and is_tap_major(t) for t in tuple(wx) + tuple(wh))):
            # tap-major master weights with slots in the gradient sink: the partial sums of the two fused launches are summed
            # straight into the gates' slots by the sink's one reduce launch — live taps only, no [4C, C, 3, 3] intermediates
            gwx = _fused_gates_to_sink(sink, wx, x5, gpre[0], w4shape)
            hin = hs[:steps - 1].reshape((steps - 1) * B, Cc, 1, h, w)    # time folded into the batch
            gin = gpre[1:].reshape((steps - 1) * B, 4 * Cc, 1, h, w)
            gwh = _fused_gates_to_sink(sink, wh, hin, gin, w4shape)
            return (gx, None) + tuple(gwx) + tuple(gbx) + tuple(gwh)
        gwx4 = conv_wgrad_raw(x5, gpre[0], w4shape)                       # all four gates at once
        if steps > 1:
            hin = hs[:steps - 1].reshape((steps - 1) * B, Cc, 1, h, w)    # time folded into the batch
            gin = gpre[1:].reshape((steps - 1) * B, 4 * Cc, 1, h, w)
            gwh4 = conv_wgrad_raw(hin, gin, w4shape)
        else:
            gwh4 = _zeros_like(gwx4)
        gwx = [gwx4[g * Cc:(g + 1) * Cc].squeeze(2) for g in range(4)]
        gwh = [gwh4[g * Cc:(g + 1) * Cc].squeeze(2) for g in range(4)]
        return (gx, None) + tuple(gwx) + tuple(gbx) + tuple(gwh)


def conv_lstm(x, steps, wx, bx, wh):
    return ConvLSTMFn.apply(x, steps, *(list(wx) + list(bx) + list(wh)))


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------

class RSGan(Function):
    """mean softplus(-(a - b)) == BCEWithLogits(a - b, ones)  (losses.py:79-85)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        check(lib().t2v_rsgan(_p(a), _p(b), _p(out), a.numel(), _stream()), 't2v_rsgan')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        check(lib().t2v_rsgan_bwd(_p(a), _p(b), _p(g), _p(ga), _p(gb), a.numel(), _stream()), 't2v_rsgan_bwd')
        return ga, gb


def rsgan(a, b):
    return RSGan.apply(a, b)


class RSGanMeanG(Function):
    """mean over n pairs (a_l, b_l) of mean softplus(-(a_l - b_l)): the relativistic loss of every pyramid level and their mean
    (cond_gan.py:121-154) in one launch; bit-identical to `rsgan` per level + `scalar_mean`. args: n, a_0.., b_0.."""

    @staticmethod
    def forward(ctx, n, *ts):
        from ._lib import MultiJob
        a, b = [_c(t) for t in ts[:n]], [_c(t) for t in ts[n:]]
        if any(x.numel() != y.numel() for x, y in zip(a, b)):
            raise ValueError('rsgan_mean_levels: logit vectors of a level differ in size')
        arr = (MultiJob * n)()
        for q, x, y in zip(arr, a, b):
            q.a, q.b, q.n = x.data_ptr(), y.data_ptr(), x.numel()
        out = torch.empty((), device=a[0].device, dtype=torch.float32)
        check(lib().t2v_rsgan_mean_multi(arr, n, _p(out), _stream()), 't2v_rsgan_mean_multi')
        ctx.save_for_backward(*a, *b)
        ctx.n = n
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from ._lib import MultiJob
        n = ctx.n
        saved = ctx.saved_tensors
        a, b = saved[:n], saved[n:]
        g = _c(g)
        ga = [torch.empty_like(x) if ctx.needs_input_grad[1 + i] else None for i, x in enumerate(a)]
        gb = [torch.empty_like(y) if ctx.needs_input_grad[1 + n + i] else None for i, y in enumerate(b)]
        arr = (MultiJob * n)()
        for q, x, y, u, v in zip(arr, a, b, ga, gb):
            q.a, q.b, q.n = x.data_ptr(), y.data_ptr(), x.numel()
            q.out = u.data_ptr() if u is not None else None
            q.out2 = v.data_ptr() if v is not None else None
        check(lib().t2v_rsgan_mean_multi_bwd(arr, n, _p(g), _stream()), 't2v_rsgan_mean_multi_bwd')
        return (None,) + tuple(ga) + tuple(gb)


def rsgan_mean_levels(as_, bs):
    as_, bs = list(as_), list(bs)
    return RSGanMeanG.apply(len(as_), *(as_ + bs))


LOSS_KINDS = {'vanilla': 1, 'hinge': 2, 'wasserstein': 3, 'rasgan': 4, 'ralsgan': 5}


class GanLoss(Function):
    """The rest of the loss zoo on D's logits (losses.py:19-68,87-133): one launch forward, one backward."""

    @staticmethod
    def forward(ctx, real, fake, kind, side, margin):
        fake = _c(fake)
        real = _c(real) if real is not None else None
        ctx.save_for_backward(real, fake)
        ctx.cfg = (kind, side, float(margin))
        out = torch.empty((), device=fake.device, dtype=torch.float32)
        check(lib().t2v_gan_loss(_p(real), _p(fake), _p(out), real.numel() if real is not None else 0, fake.numel(), kind, side,
                                 float(margin), _stream()), 't2v_gan_loss')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        real, fake = ctx.saved_tensors
        kind, side, margin = ctx.cfg
        gr = torch.empty_like(real) if (real is not None and ctx.needs_input_grad[0]) else None
        gf = torch.empty_like(fake) if ctx.needs_input_grad[1] else None
        check(lib().t2v_gan_loss_bwd(_p(real), _p(fake), _p(_c(g)), _p(gr), _p(gf), real.numel() if real is not None else 0,
                                     fake.numel(), kind, side, margin, _stream()), 't2v_gan_loss_bwd')
        return gr, gf, None, None, None


def gan_loss(kind, side, real, fake, margin=0.0):
    return GanLoss.apply(real, fake, LOSS_KINDS[kind], side, margin)


def lerp_rows(alpha, xr, xf):
    """alpha[b]*xr[b] + (1-alpha[b])*xf[b]  (losses.py:146); no autograd (inputs are detached)."""
    xr, xf, alpha = _c(xr), _c(xf), _c(alpha)
    rows = xr.shape[0]
    out = torch.empty_like(xr)
    check(lib().t2v_lerp_rows(_p(alpha), _p(xr), _p(xf), _p(out), rows, xr.numel() // rows, _stream()), 't2v_lerp_rows')
    return out


def cat_lerp_group(reals, fakes, alphas=None):
    """The discriminator step's inputs for all pyramid levels in ONE launch: `[torch.cat((r, f)) for r, f in levels]` and,
    with `alphas` (one [b] tensor per level), the gradient penalty's interpolates `a * r + (1 - a) * f` as well. No autograd:
    for detached clips only (the D step)."""
    reals, fakes = [_c(t) for t in reals], [_c(t) for t in fakes]
    if any(t.requires_grad for t in reals + fakes) and torch.is_grad_enabled():
        raise ValueError('cat_lerp_group takes detached clips')
    rfs = [torch.empty((r.shape[0] + f.shape[0],) + tuple(r.shape[1:]), device=r.device, dtype=torch.float32) for r, f in zip(reals, fakes)]
    xhs = [torch.empty_like(r) for r in reals] if alphas is not None else [None] * len(reals)
    jobs = []
    for i, (r, f) in enumerate(zip(reals, fakes)):
        if r.shape != f.shape:
            raise ValueError('real / generated clip shapes differ: %s vs %s' % (tuple(r.shape), tuple(f.shape)))
        jobs.append(dict(a=r, b=f, c=_c(alphas[i]) if alphas is not None else None, out=rfs[i], out2=xhs[i], n=r.numel(),
                         d0=r.numel() // r.shape[0]))
    _mj(MJ_CATLERP, jobs)
    return rfs, (xhs if alphas is not None else None)


class RowSqNorm(Function):
    """[b, ...] -> [b]: squared L2 norm per sample (losses.py:182)."""

    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        rows = g.shape[0]
        ctx.save_for_backward(g)
        out = torch.empty((rows,), device=g.device, dtype=torch.float32)
        check(lib().t2v_row_sqnorm(_p(g), _p(out), rows, g.numel() // rows, _stream()), 't2v_row_sqnorm')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        g, = ctx.saved_tensors
        go = _c(go)
        rows = g.shape[0]
        out = torch.empty_like(g)
        check(lib().t2v_row_scale(_p(go), 2.0, _p(g), _p(out), rows, g.numel() // rows, _stream()), 't2v_row_scale')
        return out


def row_sqnorm(g):
    return RowSqNorm.apply(g)


def pyramid_gather(x, Bo, To, Ho, Wo, sb, st, bt, bt_dev=None):
    """y[b,c,t,h,w] = x[b*sb, c, t*st+bt, nearest(h), nearest(w)] — Subsample + F.interpolate(nearest).
    bt_dev: optional int32 device tensor holding the phase (graph replay)."""
    x = _c(x)
    B, Cc, T, H, W = x.shape
    y = torch.empty((Bo, Cc, To, Ho, Wo), device=x.device, dtype=torch.float32)
    check(lib().t2v_pyramid_gather(_p(x), _p(y), B, Cc, T, H, W, Bo, To, Ho, Wo, sb, st, bt, _p(bt_dev), _stream()),
          't2v_pyramid_gather')
    return y


class PyramidGather(Function):
    """Differentiable batch/time sub-sampling `x[::2, :, bt::2]` (layers.py:110) for the generator."""

    @staticmethod
    def forward(ctx, x, bt):
        B, Cc, T, H, W = x.shape
        ctx.cfg = (tuple(x.shape), bt)
        return pyramid_gather(x, (B + 1) // 2, (T - bt + 1) // 2, H, W, 2, 2, bt)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        shp, bt = ctx.cfg
        g = _c(g)
        gx = torch.empty(shp, device=g.device, dtype=torch.float32)
        check(lib().t2v_fill(_p(gx), 0.0, gx.numel(), _stream()), 't2v_fill')
        B, Cc, T, H, W = shp
        check(lib().t2v_pyramid_scatter(_p(g), _p(gx), B, Cc, T, H * W, g.shape[0], g.shape[2], 2, 2, bt, _stream()),
              't2v_pyramid_scatter')
        return gx, None


def adam_step(p, g, m, v, lr, b1, b2, eps, step, gscale=1.0, step_dev=None):
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    check(lib().t2v_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, bc1, bc2, gscale, _p(step_dev), _stream()),
          't2v_adam')


def adam_step_multi(items, lr, b1, b2, eps, step, gscale=1.0, step_dev=None):
    """`items`: [(p, g, exp_avg, exp_avg_sq)] that share `step` — one launch per 64 tensors (`t2v_adam_multi`)."""
    from ._lib import AdamJob
    arr = (AdamJob * len(items))()
    for a, (p, g, m, v) in zip(arr, items):
        a.p, a.g, a.m, a.v, a.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    check(lib().t2v_adam_multi(arr, len(items), lr, b1, b2, eps, bc1, bc2, gscale, _p(step_dev), _stream()), 't2v_adam_multi')


# ------------------------------------------------------------------------------------------------
# layout glue (strided copies done by t2v_pyramid_gather / t2v_copy2d — no ATen kernels)
# ------------------------------------------------------------------------------------------------

def _copy2d(src, src_off, src_ld, dst, dst_off, dst_ld, rows, cols):
    check(lib().t2v_copy2d(C.c_void_p(src.data_ptr() + 4 * src_off), src_ld, C.c_void_p(dst.data_ptr() + 4 * dst_off), dst_ld,
                           rows, cols, _stream()), 't2v_copy2d')


class SliceCols(Function):
    """x[:, off:off+n] of a 2-D tensor (adjoint: EmbedCols)."""

    @staticmethod
    def forward(ctx, x, off, n):
        x = _c(x)
        ctx.cfg = (off, x.shape[1])
        out = torch.empty((x.shape[0], n), device=x.device, dtype=torch.float32)
        _copy2d(x, off, x.shape[1], out, 0, n, x.shape[0], n)
        return out

    @staticmethod
    def backward(ctx, g):
        off, total = ctx.cfg
        return EmbedCols.apply(g, off, total), None, None


class EmbedCols(Function):
    @staticmethod
    def forward(ctx, g, off, total):
        g = _c(g)
        ctx.cfg = (off, g.shape[1])
        out = torch.empty((g.shape[0], total), device=g.device, dtype=torch.float32)
        check(lib().t2v_fill(_p(out), 0.0, out.numel(), _stream()), 't2v_fill')
        _copy2d(g, 0, g.shape[1], out, off, total, g.shape[0], g.shape[1])
        return out

    @staticmethod
    def backward(ctx, gg):
        off, n = ctx.cfg
        return SliceCols.apply(gg, off, n), None, None


class CatFeatures(Function):
    """torch.cat((a, b), dim=1) for 2-D tensors (resnet3d.py:53, tganv2_cond/gen.py:68)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        na, nb = a.shape[1], b.shape[1]
        ctx.cfg = (na, nb)
        out = torch.empty((a.shape[0], na + nb), device=a.device, dtype=torch.float32)
        _copy2d(a, 0, na, out, 0, na + nb, a.shape[0], na)
        _copy2d(b, 0, nb, out, na, na + nb, a.shape[0], nb)
        return out

    @staticmethod
    def backward(ctx, g):
        na, nb = ctx.cfg
        ga = SliceCols.apply(g, 0, na) if ctx.needs_input_grad[0] else None
        gb = SliceCols.apply(g, na, nb) if ctx.needs_input_grad[1] else None
        return ga, gb


def cat_features(a, b):
    return CatFeatures.apply(a, b)


def _permute_01(x, A, Bd, inner):
    """[A,B,inner] -> [B,A,inner] dense copy."""
    out = torch.empty((Bd, A, inner), device=x.device, dtype=torch.float32)
    check(lib().t2v_permute01(_p(x), _p(out), A, Bd, inner, _stream()), 't2v_permute01')
    return out


class Permute01(Function):
    """Swap the two leading axes of a contiguous tensor viewed as [A,B,inner]."""

    @staticmethod
    def forward(ctx, x, A, Bd):
        x = _c(x)
        inner = x.numel() // (A * Bd)
        ctx.cfg = (A, Bd, tuple(x.shape))
        return _permute_01(x, A, Bd, inner)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        A, Bd, shp = ctx.cfg
        g = _c(g)
        return _permute_01(g, Bd, A, g.numel() // (A * Bd)).view(shp), None, None


def time_to_batch(hs):
    """[T,B,C,h,w] -> [B*T,C,h,w]  (torch.stack(x).permute(1,0,2,3,4) + merge_frames, gen.py:75-96)."""
    T, B = hs.shape[0], hs.shape[1]
    return Permute01.apply(hs, T, B).view((B * T,) + tuple(hs.shape[2:]))


def frames_to_video(r, T):
    """[b*T,C,H,W] -> [b,C,T,H,W]  (split_frames + time_first, gen.py:117-118). With one channel (or one frame) the two layouts
    are the same memory: a view, no kernel."""
    bT, Cc, H, W = r.shape
    if (Cc == 1 or T == 1) and r.is_contiguous():
        return r.view(bT // T, Cc, T, H, W)
    return _FramesToVideo.apply(r, T)


class _FramesToVideo(Function):
    @staticmethod
    def forward(ctx, r, T):
        r = _c(r)
        bT, Cc, H, W = r.shape
        b = bT // T
        ctx.cfg = (b, T, Cc, H, W)
        out = torch.empty((b, Cc, T, H, W), device=r.device, dtype=torch.float32)
        check(lib().t2v_permute12(_p(r), _p(out), b, T, Cc, H * W, _stream()), 't2v_permute12')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        b, T, Cc, H, W = ctx.cfg
        g = _c(g)
        out = torch.empty((b * T, Cc, H, W), device=g.device, dtype=torch.float32)
        check(lib().t2v_permute12(_p(g), _p(out), b, Cc, T, H * W, _stream()), 't2v_permute12')
        return out, None


class _SubsampleFrames(Function):
    """[b*T,C,h,w] -> keep samples ::2 and frames bt::2 -> [ceil(b/2)*(T/2),C,h,w] (gen.py:98-109)."""

    @staticmethod
    def forward(ctx, x, T, bt, bt_dev):
        x = _c(x)
        bT, Cc, H, W = x.shape
        b = bT // T
        if bt_dev is not None and T % 2:
            raise ValueError('a device-resident phase needs an even frame count (the output shape must not depend on it)')
        bo, To = (b + 1) // 2, (T - bt + 1) // 2
        ctx.cfg = (tuple(x.shape), b, T, bt, bo, To)
        ctx.bt_dev = bt_dev
        out = torch.empty((bo * To, Cc, H, W), device=x.device, dtype=torch.float32)
        check(lib().t2v_subsample_frames(_p(x), _p(out), b, T, Cc * H * W, bo, To, bt, 0, _p(bt_dev), _stream()),
              't2v_subsample_frames')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        shp, b, T, bt, bo, To = ctx.cfg
        g = _c(g)
        gx = torch.empty(shp, device=g.device, dtype=torch.float32)
        check(lib().t2v_fill(_p(gx), 0.0, gx.numel(), _stream()), 't2v_fill')
        check(lib().t2v_subsample_frames(_p(gx), _p(g), b, T, shp[1] * shp[2] * shp[3], bo, To, bt, 1, _p(ctx.bt_dev), _stream()),
              't2v_subsample_frames(adjoint)')
        return gx, None, None, None


def subsample_frames(x, T, bt, bt_dev=None):
    return _SubsampleFrames.apply(x, T, bt, bt_dev)


# ------------------------------------------------------------------------------------------------
# scalar glue: combining per-level / per-head losses (cond_gan.py:51-61,106-118, losses.py:207)
# ------------------------------------------------------------------------------------------------

def ones_like(t):
    out = torch.empty_like(t)
    check(lib().t2v_fill(_p(out), 1.0, out.numel(), _stream()), 't2v_fill')
    return out


_ones = {}


def ones_cached(t):
    """A constant all-ones tensor of t's shape (the roots of the gradient penalty's `autograd.grad`): filled once, never written
    again, so every later iteration saves the fill launch."""
    key = (tuple(t.shape), t.device)
    o = _ones.get(key)
    if o is None:
        o = _ones[key] = ones_like(t.detach())
    return o


class ScalarCombine(Function):
    """sum_i w_i * s_i over 0-d device tensors (one kernel; torch.stack(...).mean()/sum() in the reference)."""

    @staticmethod
    def forward(ctx, weights, *scalars):
        n = len(scalars)
        ctx.weights = weights
        ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in scalars])
        w = (C.c_float * n)(*weights)
        out = torch.empty((), device=scalars[0].device, dtype=torch.float32)
        check(lib().t2v_scalar_combine(ptrs, w, n, _p(out), _stream()), 't2v_scalar_combine')
        return out

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g if w == 1.0 else ScaleConst.apply(g, w) for w in ctx.weights)     # (weight 1: no launch)


class ScaleConst(Function):
    """y = c * a for a host constant c."""

    @staticmethod
    def forward(ctx, a, c):
        a = _c(a)
        ctx.c = c
        out = torch.empty_like(a)
        check(lib().t2v_axpby(c, _p(a), 0.0, None, _p(out), a.numel(), _stream()), 't2v_axpby')
        return out

    @staticmethod
    def backward(ctx, g):
        return ScaleConst.apply(g, ctx.c), None


def scalar_sum(scalars, weights=None):
    scalars = list(scalars)
    if weights is None:
        weights = [1.0] * len(scalars)
    return ScalarCombine.apply(tuple(float(w) for w in weights), *scalars)


def scalar_mean(scalars):
    scalars = list(scalars)
    return scalar_sum(scalars, [1.0 / len(scalars)] * len(scalars))


class VecSum(Function):
    """scale * sum(v) for a small 1-D device vector -> 0-d (losses.py:203 combine=torch.sum)."""

    @staticmethod
    def forward(ctx, v, scale):
        v = _c(v)
        ctx.n, ctx.scale = v.numel(), scale
        out = torch.empty((), device=v.device, dtype=torch.float32)
        check(lib().t2v_rowsum(_p(v), _p(out), 1, v.numel(), _stream()), 't2v_rowsum')
        if scale != 1.0:
            check(lib().t2v_axpby(scale, _p(out), 0.0, None, _p(out), 1, _stream()), 't2v_axpby')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        out = torch.empty((ctx.n,), device=g.device, dtype=torch.float32)
        gg = _c(g).view(1)
        check(lib().t2v_rowbcast(_p(gg), _p(out), 1, ctx.n, _stream()), 't2v_rowbcast')
        if ctx.scale != 1.0:
            check(lib().t2v_axpby(ctx.scale, _p(out), 0.0, None, _p(out), ctx.n, _stream()), 't2v_axpby')
        return out, None


def vec_sum(v, scale=1.0):
    return VecSum.apply(v, float(scale))


class GatherRows(Function):
    """x[perm] for a 2-D tensor (mismatched captions, cond_gan.py:133)."""

    @staticmethod
    def forward(ctx, x, perm_dev, inverse=False):
        x = _c(x)
        ctx.save_for_backward(perm_dev)
        ctx.inverse = inverse
        out = torch.empty_like(x)
        check(lib().t2v_gather_rows(_p(x), _p(perm_dev), _p(out), x.shape[0], x.shape[1], int(inverse), _stream()), 't2v_gather_rows')
        return out

    @staticmethod
    def backward(ctx, g):
        perm_dev, = ctx.saved_tensors
        return GatherRows.apply(g, perm_dev, not ctx.inverse), None, None


def gather_rows(x, perm):
    import numpy as _np
    perm_dev = torch.from_numpy(_np.ascontiguousarray(perm, dtype=_np.int32)).to(x.device)
    return GatherRows.apply(x, perm_dev, False)


_transposed_cache = {}


def transposed_weight(w):
    """w^T of a 2-D parameter as a dense tensor, cached until the parameter changes (same tags as the packed conv weights)."""
    tag = _wtag(w)
    hit = _transposed_cache.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == tag:
        return hit[2]
    wt = _permute_01(_c(w.detach()), w.shape[0], w.shape[1], 1).view(w.shape[1], w.shape[0])
    _transposed_cache[id(w)] = (weakref.ref(w), tag, wt)
    return wt


def lstm_encode(tokens, lengths, embed_weight, lstm, hidden_size, num_layers, bidirectional):
    """Packed-sequence (Bi-)LSTM forward of the sentence encoder on the HIP kernels (models/txt/basic.py:49-70), no autograd
    (the GAN loop detaches the sentence code unless --end2end). tokens [B,L] int64 (sorted by length, desc), lengths: list —
    or, for graph capture, int32 device tokens [B,L] and lengths = (L, int32 device tensor [B]).
    `lstm` is the nn.LSTM that holds the parameters (state_dict layout of the reference). Returns
    (out [B,L,D*H] zero beyond each length, (h_n, c_n) [layers*D,B,H]) like nn.LSTM on a packed batch."""
    B, L = int(tokens.shape[0]), int(lengths[0])
    dev = embed_weight.device
    H, D = hidden_size, (2 if bidirectional else 1)
    with torch.no_grad():
        if isinstance(lengths, tuple) and len(lengths) == 2 and isinstance(lengths[1], torch.Tensor):
            # (L, int32 device tensor of the B lengths) + int32 device tokens [B, L] dense: nothing is uploaded here, so the
            # whole encoder can be captured in a HIP graph (gan.trainer.GraphedSentenceEncoder)
            len_dev = lengths[1]
            tok = tokens.view(-1)
            if tokens.dtype != torch.int32 or not tokens.is_cuda or tokens.shape[1] != L or not tokens.is_contiguous():
                raise TypeError('device-resident encoding needs dense int32 device tokens [B, L]')
        else:
            tok = tokens[:, :L].to(device=dev, dtype=torch.int32).contiguous().view(-1)
            len_dev = torch.tensor([int(l) for l in lengths], dtype=torch.int32).to(dev)
        E = embed_weight.shape[1]
        x = torch.empty((B * L, E), device=dev, dtype=torch.float32)
        check(lib().t2v_gather_rows(_p(_c(embed_weight.detach())), _p(tok), _p(x), B * L, E, 0, _stream()), 't2v_gather_rows')
        h_n = torch.empty((num_layers * D, B, H), device=dev, dtype=torch.float32)
        c_n = torch.empty((num_layers * D, B, H), device=dev, dtype=torch.float32)
        zero = _zeros_like(h_n[0])
        inp = x                                                   # [B*L, In]
        for layer in range(num_layers):
            out = torch.empty((B, L, D * H), device=dev, dtype=torch.float32)
            xprojs, whts, bufss = [], [], []
            for d in range(D):
                sfx = '_l%d%s' % (layer, '_reverse' if d else '')
                w_ih, w_hh = getattr(lstm, 'weight_ih' + sfx), getattr(lstm, 'weight_hh' + sfx)     # parameters: packed / transposed once
                bsum = _ew(lib().t2v_add, 't2v_add', getattr(lstm, 'bias_ih' + sfx).detach(), getattr(lstm, 'bias_hh' + sfx).detach())
                xprojs.append(conv_fwd_raw(_as5(inp), _as5(w_ih), bsum).view(B, L, 4 * H))        # all time steps: one GEMM
                whts.append(transposed_weight(w_hh))                                              # [H, 4H]: coalesced reads in the step kernel
                bufss.append([(torch.empty_like(zero), torch.empty_like(zero)) for _ in range(2)])     # ping-pong (h, c)

            def step_args(d, step):
                t = L - 1 - step if d else step
                hp, cp = (zero, zero) if step == 0 else bufss[d][(step - 1) & 1]
                hn_, cn_ = (h_n[layer * D + d], c_n[layer * D + d]) if step == L - 1 else bufss[d][step & 1]
                xp = xprojs[d].data_ptr() + 4 * t * 4 * H
                op = out.data_ptr() + 4 * (t * D * H + d * H)
                return t, [xp, whts[d].data_ptr(), hp.data_ptr(), cp.data_ptr(), hn_.data_ptr(), cn_.data_ptr(), op]
            for step in range(L):
                if D == 2:            # the two directions of a step are independent: one launch
                    (t0, p0), (t1, p1) = step_args(0, step), step_args(1, step)
                    check(lib().t2v_lstm_seq_step2((C.c_void_p * 14)(*(p0 + p1)), (C.c_int32 * 2)(t0, t1), L * 4 * H, L * D * H,
                                                   _p(len_dev), B, H, _stream()), 't2v_lstm_seq_step2')
                else:
                    t, q = step_args(0, step)
                    check(lib().t2v_lstm_seq_step(C.c_void_p(q[0]), L * 4 * H, C.c_void_p(q[1]), C.c_void_p(q[2]), C.c_void_p(q[3]),
                                                  C.c_void_p(q[4]), C.c_void_p(q[5]), C.c_void_p(q[6]), L * D * H, _p(len_dev), t, B, H,
                                                  _stream()), 't2v_lstm_seq_step')
            inp = out.view(B * L, D * H)
        return out, (h_n, c_n)


# ------------------------------------------------------------------------------------------------
# text-encoder pre-training (train/txt.py:160-178): differentiable LSTM direction, embedding, cross entropy, arg-max
# ------------------------------------------------------------------------------------------------

def _ptr(t, off_floats=0):
    return C.c_void_p(t.data_ptr() + 4 * off_floats)


class LstmDirFn(Function):
    """One direction of one nn.LSTM layer over a padded batch with pack_padded_sequence semantics (samples past their length keep
    their state and emit zeros). xproj [B,L,4H] = x W_ih^T + b_ih + b_hh for all time steps (one GEMM, outside), w_hh [4H,H],
    h0 / c0 [B,H]. Returns (out [B,L,H], h_n, c_n). One small launch per time step in each sweep; the recurrent weight gradient
    is one GEMM over all steps."""

    @staticmethod
    def forward(ctx, xproj, w_hh, h0, c0, len_dev, reverse):
        xproj, w_hh, h0, c0 = _c(xproj), _c(w_hh), _c(h0), _c(c0)
        B, L, H4 = xproj.shape
        H = H4 // 4
        dev = xproj.device
        hprev = torch.empty((B, L, H), device=dev, dtype=torch.float32)      # state ENTERING time step t
        cprev = torch.empty((B, L, H), device=dev, dtype=torch.float32)
        gates = torch.empty((B, L, H4), device=dev, dtype=torch.float32)
        out = torch.empty((B, L, H), device=dev, dtype=torch.float32)
        h_n = torch.empty((B, H), device=dev, dtype=torch.float32)
        c_n = torch.empty((B, H), device=dev, dtype=torch.float32)
        order = list(range(L - 1, -1, -1)) if reverse else list(range(L))
        # [H, 4H] for the step kernel's coalesced reads (cached per parameter: the decoder calls this once per position)
        w_hh_t = transposed_weight(w_hh) if isinstance(w_hh, torch.nn.Parameter) else _permute_01(w_hh, H4, H, 1)
        _copy2d(h0, 0, H, hprev, order[0] * H, L * H, B, H)
        _copy2d(c0, 0, H, cprev, order[0] * H, L * H, B, H)
        for s_, t in enumerate(order):
            last = s_ == L - 1
            tn = t if last else order[s_ + 1]
            check(lib().t2v_lstm_train_step(_ptr(xproj, t * H4), L * H4, _p(w_hh_t), _ptr(hprev, t * H), L * H, _ptr(cprev, t * H), L * H,
                                            _p(h_n) if last else _ptr(hprev, tn * H), H if last else L * H,
                                            _p(c_n) if last else _ptr(cprev, tn * H), H if last else L * H,
                                            _ptr(out, t * H), L * H, _ptr(gates, t * H4), L * H4, _p(len_dev), t, B, H, _stream()),
                  't2v_lstm_train_step')
        ctx.save_for_backward(w_hh, hprev, cprev, gates, c_n, len_dev)
        ctx.order = order
        return out, h_n, c_n

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, d_hn, d_cn):
        w_hh, hprev, cprev, gates, c_n, len_dev = ctx.saved_tensors
        order = ctx.order
        B, L, H = hprev.shape
        H4 = 4 * H
        dev = hprev.device
        DH = _c(d_hn).clone() if d_hn is not None else _zeros_like(c_n)
        DC = _c(d_cn).clone() if d_cn is not None else _zeros_like(c_n)
        d_out = _c(d_out) if d_out is not None else None
        dG = torch.empty((B, L, H4), device=dev, dtype=torch.float32)
        for s_ in range(L - 1, -1, -1):
            t = order[s_]
            first = s_ == L - 1
            tl = t if first else order[s_ + 1]
            check(lib().t2v_lstm_train_step_bwd(_ptr(d_out, t * H) if d_out is not None else None, L * H,
                                                None if first else _ptr(dG, tl * H4), L * H4, tl, _p(w_hh), _p(DH), _p(DC),
                                                _ptr(gates, t * H4), L * H4, _ptr(cprev, t * H), L * H,
                                                _p(c_n) if first else _ptr(cprev, tl * H), H if first else L * H,
                                                _ptr(dG, t * H4), L * H4, _p(len_dev), t, B, H, int(first), 0, _stream()),
                  't2v_lstm_train_step_bwd')
        dw = dh0 = dc0 = None
        if ctx.needs_input_grad[1]:
            dw = _bmm_raw(dG.view(1, B * L, H4), hprev.view(1, B * L, H), H4, H, B * L, True, False).view(H4, H)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            t0 = order[0]            # finish DH for the state entering the first step; DC already is dL/dc0
            check(lib().t2v_lstm_train_step_bwd(None, 0, _ptr(dG, t0 * H4), L * H4, t0, _p(w_hh), _p(DH), _p(DC), None, 0, None, 0,
                                                None, 0, None, 0, _p(len_dev), 0, B, H, 0, 1, _stream()), 't2v_lstm_train_step_bwd')
            dh0, dc0 = DH, DC
        return dG, dw, dh0, dc0, None, None


def lstm_direction(xproj, w_hh, h0, c0, len_dev, reverse=False):
    return LstmDirFn.apply(xproj, w_hh, h0, c0, len_dev, bool(reverse))


class EmbeddingFn(Function):
    """nn.Embedding lookup: weight[tokens] -> [N,E]; the adjoint adds rows in token order without atomics."""

    @staticmethod
    def forward(ctx, weight, tok_dev):
        weight = _c(weight)
        ctx.save_for_backward(tok_dev)
        ctx.shape = tuple(weight.shape)
        N = tok_dev.numel()
        out = torch.empty((N, weight.shape[1]), device=weight.device, dtype=torch.float32)
        check(lib().t2v_gather_rows(_p(weight), _p(tok_dev), _p(out), N, weight.shape[1], 0, _stream()), 't2v_gather_rows')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        tok_dev, = ctx.saved_tensors
        V, E = ctx.shape
        dW = torch.empty((V, E), device=g.device, dtype=torch.float32)
        check(lib().t2v_fill(_p(dW), 0.0, dW.numel(), _stream()), 't2v_fill')
        check(lib().t2v_embedding_bwd(_p(_c(g)), _p(tok_dev), _p(dW), tok_dev.numel(), E, V, _stream()), 't2v_embedding_bwd')
        return dW, None


def embedding(weight, tokens):
    """tokens: integer tensor of any shape (host or device) -> [tokens.numel(), E]."""
    tok = tokens.reshape(-1).to(device=weight.device, dtype=torch.int32).contiguous()
    return EmbeddingFn.apply(weight, tok)


class XentRows(Function):
    """Per-row cross entropy of logits [N,V] against int targets [N] (nn.CrossEntropyLoss before its reduction)."""

    @staticmethod
    def forward(ctx, logits, tgt_dev):
        logits = _c(logits)
        N, V = logits.shape
        loss = torch.empty((N,), device=logits.device, dtype=torch.float32)
        lse = torch.empty((N,), device=logits.device, dtype=torch.float32)
        check(lib().t2v_xent_fwd(_p(logits), _p(tgt_dev), _p(loss), _p(lse), N, V, _stream()), 't2v_xent_fwd')
        ctx.save_for_backward(logits, tgt_dev, lse)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logits, tgt_dev, lse = ctx.saved_tensors
        N, V = logits.shape
        dx = torch.empty_like(logits)
        check(lib().t2v_xent_bwd(_p(logits), _p(tgt_dev), _p(lse), _p(_c(g)), _p(dx), N, V, _stream()), 't2v_xent_bwd')
        return dx, None


def cross_entropy(logits, targets, reduction='mean'):
    """nn.CrossEntropyLoss()(logits [N,V], targets [N]) with reduction 'mean' (train/txt.py:158) or 'sum' (:55)."""
    tgt = targets.reshape(-1).to(device=logits.device, dtype=torch.int32).contiguous()
    rows = XentRows.apply(logits, tgt)
    return vec_sum(rows, 1.0 / rows.numel() if reduction == 'mean' else 1.0)


def argmax_rows(logits):
    """logits [N,V] -> int64 [N] (first index of each row's maximum)."""
    logits = _c(logits.detach())
    idx = torch.empty((logits.shape[0],), device=logits.device, dtype=torch.int32)
    check(lib().t2v_argmax_rows(_p(logits), _p(idx), logits.shape[0], logits.shape[1], _stream()), 't2v_argmax_rows')
    return idx.to(torch.int64)


class StackSteps(Function):
    """torch.stack(steps, 1) for T tensors [B,V] -> [B,T,V] (basic.py:98)."""

    @staticmethod
    def forward(ctx, *steps):
        T = len(steps)
        B, V = steps[0].shape
        out = torch.empty((B, T, V), device=steps[0].device, dtype=torch.float32)
        for i, st in enumerate(steps):
            _copy2d(_c(st), 0, V, out, i * V, T * V, B, V)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, V = g.shape
        g2 = g.reshape(B, T * V)
        return tuple(SliceCols.apply(g2, i * V, V) if ctx.needs_input_grad[i] else None for i in range(T))


def stack_steps(steps):
    return StackSteps.apply(*steps)


def lstm_stack(x, lengths_dev, lstm, hidden_size, num_layers, bidirectional, initial_state=None):
    """Differentiable multi-layer (bi-)LSTM over a padded batch: x [B,L,In] -> (out [B,L,D*H], (h_n, c_n) [layers*D,B,H]) like
    nn.LSTM on a packed batch (lengths sorted descending). `lstm` is the nn.LSTM holding the parameters."""
    B, L = int(x.shape[0]), int(x.shape[1])
    H, D = hidden_size, (2 if bidirectional else 1)
    if initial_state is not None:
        for part in initial_state:
            if len(part) != num_layers * D or any(tuple(t.shape) != (B, H) for t in part):
                raise ValueError('initial state must be %d tensors of shape [%d, %d] per (h, c); got %s'
                                 % (num_layers * D, B, H, [tuple(t.shape) for t in part]))
    inp = x.reshape(B * L, -1)
    hs, cs = [], []
    zero = None
    for layer in range(num_layers):
        outs = []
        for d in range(D):
            sfx = '_l%d%s' % (layer, '_reverse' if d else '')
            bias = add(getattr(lstm, 'bias_ih' + sfx), getattr(lstm, 'bias_hh' + sfx))
            xproj = linear(inp, getattr(lstm, 'weight_ih' + sfx), bias).view(B, L, 4 * H)
            if initial_state is None:
                if zero is None:
                    zero = torch.empty((B, H), device=x.device, dtype=torch.float32)
                    check(lib().t2v_fill(_p(zero), 0.0, zero.numel(), _stream()), 't2v_fill')
                h0 = c0 = zero
            else:
                h0, c0 = initial_state[0][layer * D + d], initial_state[1][layer * D + d]
            o, h_n, c_n = lstm_direction(xproj, getattr(lstm, 'weight_hh' + sfx), h0, c0, lengths_dev, reverse=bool(d))
            outs.append(o.view(B * L, H))
            hs.append(h_n)
            cs.append(c_n)
        inp = outs[0] if D == 1 else cat_features(outs[0], outs[1])
    return inp.view(B, L, D * H), (hs, cs)


def head_rows(x, n):
    """x[0:n] — a leading-rows slice of a contiguous tensor is a view (no kernel, no copy)."""
    return x[0:n]


def stride_rows(x, step):
    """x[::step] as a dense tensor (cond[::2], trainer.py:160)."""
    x = _c(x)
    rows = (x.shape[0] + step - 1) // step
    cols = x.numel() // x.shape[0]
    out = torch.empty((rows,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    _copy2d(x, 0, cols * step, out, 0, cols, rows, cols)
    return out


def video_to_channel_first(x):
    """[B,T,C,H,W] -> [B,C,T,H,W] dense (x.permute(0,2,1,3,4), trainer.py:204)."""
    x = _c(x)
    B, T, Cc, H, W = x.shape
    if Cc == 1 or T == 1:
        return x.view(B, Cc, T, H, W)
    out = torch.empty((B, Cc, T, H, W), device=x.device, dtype=torch.float32)
    check(lib().t2v_permute12(_p(x), _p(out), B, T, Cc, H * W, _stream()), 't2v_permute12')
    return out


def copy_into(src, dst):
    """dst <- src (dense, same numel) on the copy kernel (gradient arena gather, txt2vid_amd.dist)."""
    src = _c(src)
    n = src.numel()
    _copy2d(src, 0, n, dst, 0, n, 1, n)


class CatBatch(Function):
    """torch.cat((a, b), dim=0) of two dense tensors (two copy launches into one buffer)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        ctx.na = a.shape[0]
        out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), device=a.device, dtype=torch.float32)
        na, nb = a.numel(), b.numel()
        _copy2d(a, 0, na, out, 0, na, 1, na)
        _copy2d(b, 0, nb, out, na, nb, 1, nb)
        return out

    @staticmethod
    def backward(ctx, g):
        return (g[:ctx.na] if ctx.needs_input_grad[0] else None), (g[ctx.na:] if ctx.needs_input_grad[1] else None)


def cat_batch(a, b):
    return CatBatch.apply(a, b)


class SplitRows(Function):
    """(x[:n], x[n:]) of a dense tensor as views; the adjoint writes both gradients into ONE buffer (the native slices cost a
    zero-fill + copy each and an accumulation: five launches per split in the backward)."""

    @staticmethod
    def forward(ctx, x, n):
        x = _c(x)
        ctx.n, ctx.shape = int(n), tuple(x.shape)
        ctx.set_materialize_grads(False)
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        if ga is not None and gb is not None:
            return CatBatch.apply(ga, gb), None
        out = torch.empty(ctx.shape, device=(ga if ga is not None else gb).device, dtype=torch.float32)
        check(lib().t2v_fill(_p(out), 0.0, out.numel(), _stream()), 't2v_fill')
        row = out.numel() // ctx.shape[0]
        if ga is not None:
            ga = _c(ga)
            _copy2d(ga, 0, ga.numel(), out, 0, ga.numel(), 1, ga.numel())
        else:
            gb = _c(gb)
            _copy2d(gb, 0, gb.numel(), out, ctx.n * row, gb.numel(), 1, gb.numel())
        return out, None


def split_rows(x, n):
    """(x[:n], x[n:]) with a fused adjoint."""
    return SplitRows.apply(x, n)


def tail_rows(x, n_skip):
    """x[n_skip:] — a trailing-rows slice of a contiguous tensor is a view."""
    return x[n_skip:]


# ------------------------------------------------------------------------------------------------
# random draws of one training iteration: classes in draws.py; `draws` is the instance the ops read
# ------------------------------------------------------------------------------------------------
from .draws import HostDraws, StaticDraws          # noqa: E402


draws = HostDraws()


def set_draws(d):
    global draws
    draws = d


# ------------------------------------------------------------------------------------------------
# GROUPED convolutions: the pyramid levels of the multi-scale discriminator share weights, so each layer
# is ONE launch over all levels (and one weight-gradient launch whose result is already the sum over
# levels — no per-level accumulation). Closed under differentiation like the single-tensor triple.
# ------------------------------------------------------------------------------------------------

class _TapSet(object):
    """Stands in for a geometry when packing the union of the taps a group of levels needs."""
    __slots__ = ('taps', 'taps_c', 'T', 'mask')


_tapset_cache = {}


def _tapset(T, mask):
    key = (T, mask)
    ts = _tapset_cache.get(key)
    if ts is None:
        ts = _TapSet()
        ts.taps = [t for t in range(T) if (mask >> t) & 1]
        ts.taps_c = (C.c_int32 * len(ts.taps))(*ts.taps)
        ts.T, ts.mask = T, mask
        _tapset_cache[key] = ts
    return ts


def _group_table(ins, outs, geoms, slot_of, masks=None, dstride=0):
    arr = (ConvGroup * len(ins))()
    for i, (xi, yi, g) in enumerate(zip(ins, outs, geoms)):
        a = arr[i]
        a.dstride = dstride
        a.x, a.y = xi.data_ptr(), (yi.data_ptr() if yi is not None else 0)
        a.mask = masks[i].data_ptr() if masks is not None else None
        a.N, a.D, a.H, a.W, a.ntaps = g.cg.N, g.cg.D, g.cg.H, g.cg.W, g.cg.ntaps
        for j, t in enumerate(g.taps):
            a.dz[j], a.dy[j], a.dx[j] = g.cg.dz[j], g.cg.dy[j], g.cg.dx[j]
            a.widx[j] = slot_of[t] if slot_of is not None else 0
    return arr


def conv_group_raw(xs5, w5, bias=None, relu_in=False, mode=0, masks=None, outs=None, accum=False, even_frames=False):
    """mode 0: ys[i] = conv(xs[i], w) (+bias); mode 1: data gradient (xs are dL/dy, channels swap roles).
    masks: ys[i] is zeroed where masks[i] <= 0 (the ReLU adjoint, fused into the epilogue)."""
    if len(xs5) > MAX_GROUPS:
        raise ValueError('at most %d tensors per grouped convolution' % MAX_GROUPS)
    xs5 = [_c(t) for t in xs5]
    w5 = _c(w5)
    Cout_w, Cin_w = w5.shape[0], w5.shape[1]
    cin, cout = (Cin_w, Cout_w) if mode == 0 else (Cout_w, Cin_w)
    k = tuple(w5.shape[2:])
    geoms = [conv_geom(t.shape[0], cin, t.shape[2], t.shape[3], t.shape[4], cout, k[0], k[1], k[2]) for t in xs5]
    mask = 0
    for g in geoms:
        mask |= g.mask
    ts = _tapset(geoms[0].T, mask)
    slot_of = {t: j for j, t in enumerate(ts.taps)}
    if even_frames and (mode != 0 or masks is not None or accum or outs is not None):
        raise ValueError('even_frames is a plain forward option')
    ys = outs if outs is not None else [torch.empty((t.shape[0], cout, (t.shape[2] + 1) // 2 if even_frames else t.shape[2]) + tuple(t.shape[3:]),
                                                    device=t.device, dtype=torch.float32) for t in xs5]
    if masks is not None:
        masks = [_c(m) for m in masks]
        if any(m.shape != y.shape for m, y in zip(masks, ys)):
            raise ValueError('mask / output shape mismatch')
    arr = _group_table(xs5, ys, geoms, slot_of, masks, 2 if even_frames else 0)
    n = int(lib().t2v_conv_fwd_grouped_ws_floats(arr, len(xs5), cin, cout))
    if n < 0:
        raise RuntimeError('bad grouped conv geometry')
    ws = torch.empty((n,), device=xs5[0].device, dtype=torch.float32) if n > 0 else None
    flags = ((FLAG_BIAS if bias is not None else 0) | (FLAG_RELU_IN if relu_in else 0) | (FLAG_MASK_OUT if masks is not None else 0) |
             (FLAG_ACCUM if accum else 0))
    if CONV_PRECISION == 'bf16' and lib().t2v_conv_fwd_grouped_bf16_ok(arr, len(xs5), cin, cout):
        wpb = packed_weight_bf16(w5, ts, mode)
        check(lib().t2v_conv_fwd_grouped_bf16(arr, len(xs5), cin, cout, _p(wpb), _p(bias), _p(ws), flags, _stream()),
              't2v_conv_fwd_grouped_bf16')
        return ys
    wp = packed_weight(w5, ts, mode)
    check(lib().t2v_conv_fwd_grouped(arr, len(xs5), cin, cout, _p(wp), _p(bias), _p(ws), flags, _stream()), 't2v_conv_fwd_grouped')
    return ys


def conv_group_wgrad_raw(xs5, gys5, wshape, relu_in=False, out=None, accum=False, dbias=None, accum_bias=False, even_frames=False):
    xs5, gys5 = [_c(t) for t in xs5], [_c(t) for t in gys5]
    Cout, Cin = wshape[0], wshape[1]
    k = tuple(wshape[2:])
    geoms = [conv_geom(t.shape[0], Cin, t.shape[2], t.shape[3], t.shape[4], Cout, k[0], k[1], k[2]) for t in xs5]
    arr = _group_table(xs5, gys5, geoms, None, None, 2 if even_frames else 0)        # even_frames: dL/dy on the even frames only
    query = lib().t2v_conv_wgrad_grouped_bias_slab_floats if dbias is not None else lib().t2v_conv_wgrad_grouped_slab_floats
    n = int(query(arr, len(xs5), Cin, Cout, k[0], k[1], k[2]))
    if n <= 0:
        raise RuntimeError('bad grouped wgrad geometry')
    slab = torch.empty((n,), device=xs5[0].device, dtype=torch.float32)
    dw = out if out is not None else torch.empty(tuple(wshape), device=xs5[0].device, dtype=torch.float32)
    flags = (FLAG_RELU_IN if relu_in else 0) | (FLAG_ACCUM if accum else 0) | (FLAG_BF16 if CONV_PRECISION == 'bf16' else 0)
    if dbias is not None:
        check(lib().t2v_conv_wgrad_grouped_bias(arr, len(xs5), Cin, Cout, k[0], k[1], k[2], _p(dw), _p(dbias), _p(slab),
                                                flags | (FLAG_ACCUM_BIAS if accum_bias else 0), _stream()),
              't2v_conv_wgrad_grouped_bias')
    else:
        check(lib().t2v_conv_wgrad_grouped(arr, len(xs5), Cin, Cout, k[0], k[1], k[2], _p(dw), _p(slab), flags, _stream()),
              't2v_conv_wgrad_grouped')
    return dw


def _wgrad_bias_fused(xs5, wshape):
    """True when the weight-gradient kernel for these members sums dL/dy on the side (all three kernels do)."""
    return True


def _wgrad_partial_launch(xs5, gys5, wshape, relu_in, want_bias, sink=None, even_frames=False):
    """The weight-gradient main kernel alone: returns (WgradSrc describing the k-split partial sums, the slab holding them)."""
    xs5, gys5 = [_c(t) for t in xs5], [_c(t) for t in gys5]
    Cout, Cin = wshape[0], wshape[1]
    k = tuple(wshape[2:])
    geoms = [conv_geom(t.shape[0], Cin, t.shape[2], t.shape[3], t.shape[4], Cout, k[0], k[1], k[2]) for t in xs5]
    arr = _group_table(xs5, gys5, geoms, None, None, 2 if even_frames else 0)
    query = lib().t2v_conv_wgrad_grouped_bias_slab_floats if want_bias else lib().t2v_conv_wgrad_grouped_slab_floats
    n = int(query(arr, len(xs5), Cin, Cout, k[0], k[1], k[2]))
    if n <= 0:
        raise RuntimeError('bad grouped wgrad geometry')
    slab = sink.alloc_slab(n, xs5[0].device) if sink is not None else torch.empty((n,), device=xs5[0].device, dtype=torch.float32)
    src = WgradSrc()
    flags = (FLAG_RELU_IN if relu_in else 0) | (FLAG_BF16 if CONV_PRECISION == 'bf16' else 0)
    check(lib().t2v_conv_wgrad_grouped_partial(arr, len(xs5), Cin, Cout, k[0], k[1], k[2], _p(slab), 1 if want_bias else 0, flags,
                                               C.byref(src), _side.fork(xs5, gys5) if sink is not None else _stream()),
          't2v_conv_wgrad_grouped_partial')
    return src, slab


def conv_group_wgrad_partial(xs5, gys5, wshape, relu_in, sink, wbase, dw, wacc, b, dbias, bacc, even_frames=False):
    """The weight-gradient launch WITHOUT its reduce pass: the k-split partial sums stay in a slab that `sink` keeps alive
    and sums (with every other pending slab) in its one `flush()` launch."""
    src, slab = _wgrad_partial_launch(xs5, gys5, wshape, relu_in, b is not None, sink, even_frames)
    k = tuple(wshape[2:])
    sink.add_partial(wbase, dw, wacc, b, dbias, bacc, src, slab, k[0] * k[1] * k[2], wshape[0], wshape[0] * wshape[1])


def _fused_gates_to_sink(sink, gates, x5, gy5, w4shape):
    """Weight gradients of the ConvLSTM's four gates (ONE launch on the side-by-side [4C, C, ...] weight) straight into the
    gates' tap-major sink slots: each gate is a row block of every slab plane. Returns the four gradient views."""
    Cc = w4shape[1]
    T = w4shape[2] * w4shape[3] * w4shape[4]
    src, slab = _wgrad_partial_launch([x5], [gy5], w4shape, False, False, sink)
    out = []
    for g, w in enumerate(gates):
        flat, acc = sink.take(w, deferred=True)
        sg = WgradSrc()
        C.memmove(C.byref(sg), C.byref(src), C.sizeof(WgradSrc))
        sg.slab = src.slab + 4 * g * Cc * Cc                      # rows [g*C, (g+1)*C) of each [4C][C] plane
        sink.add_partial(w, flat, acc, None, None, False, sg, slab, T, Cc, Cc * Cc, tap_major=True)
        out.append(None if acc else flat.as_strided(w.shape, w.stride()))
    return out


def _zeros_like(t):
    out = torch.empty_like(t)
    check(lib().t2v_fill(_p(out), 0.0, out.numel(), _stream()), 't2v_fill')
    return out


def zeros(shape, device, dtype=torch.float32):
    """torch.zeros on the fill kernel for fp32 device tensors (optimiser state, gradient arenas: no ATen kernel even at set-up)."""
    dev = torch.device(device)
    if dev.type != 'cuda' or dtype != torch.float32:
        return torch.zeros(shape, device=dev, dtype=dtype)
    out = torch.empty(shape, device=dev, dtype=dtype)
    if out.numel():
        check(lib().t2v_fill(_p(out), 0.0, out.numel(), _stream()), 't2v_fill')
    return out


def zeros_like(t):
    """zeros in `t`'s own memory layout (dense or tap-major): the fill kernel for fp32 device tensors."""
    if not t.is_cuda or t.dtype != torch.float32:
        return torch.zeros_like(t, memory_format=torch.preserve_format)
    out = torch.empty_like(t, memory_format=torch.preserve_format)
    if out.numel():
        check(lib().t2v_fill(_p(out), 0.0, out.numel(), _stream()), 't2v_fill')
    return out


class ConvG(Function):
    """ys = [conv(relu?(x), w) + b for x in xs] in one launch. The backward only touches the members that
    actually received a gradient AND need one: members that ride along for the forward only (detached real
    clips next to the generated ones; the gradient-penalty clips next to the real||fake batch) cost nothing."""

    @staticmethod
    def forward(ctx, w, b, relu_in, *xs):
        ctx.save_for_backward(w, *xs)
        ctx.set_materialize_grads(False)          # members nobody differentiates arrive as None, not as zeros
        ctx.has_bias, ctx.relu_in, ctx.bias = b is not None, relu_in, b
        return tuple(conv_group_raw(xs, w, b, relu_in, 0))

    @staticmethod
    def backward(ctx, *gys):
        saved = ctx.saved_tensors
        w, xs = saved[0], saved[1:]
        live = [i for i, g in enumerate(gys) if g is not None]
        gxs = [None] * len(xs)
        gw = gb = None
        if not live:
            return (None, None, None) + tuple(gxs)
        need = [i for i in live if ctx.needs_input_grad[3 + i]]
        if need:
            if ctx.relu_in:
                res = ConvDgradMaskG.apply(w, len(need), *([gys[i] for i in need] + [xs[i] for i in need]))
            else:
                res = ConvDgradG.apply(w, *[gys[i] for i in need])
            for i, r in zip(need, res):
                gxs[i] = r
        if _param_grads_enabled:
            gw, gb = _group_param_grads(w, ctx.bias if ctx.has_bias else None, ctx.relu_in, [xs[i] for i in live],
                                        [gys[i] for i in live], ctx.needs_input_grad[0], ctx.has_bias and ctx.needs_input_grad[1])
        return (gw, gb, None) + tuple(gxs)


def _group_param_grads(w, b, relu_in, lx, lg, need_w, need_b):
    """(gw, gb) of a grouped convolution over its live members: into the gradient sink when one is armed (one launch for
    both when possible), else as differentiable Functions."""
    gw = gb = None
    both = False
    if need_w and need_b:
        both, gw, gb = _to_sink_wb(w, b, lx, lg, relu_in)
    if not both:
        if need_w:
            done, gw = _to_sink_w(w, lx, lg, relu_in)
            if not done:
                gw = ConvWgradG.apply(tuple(w.shape), relu_in, len(lx), *(lx + lg))
        if need_b:
            done, gb = _to_sink(b, lambda out, acc: channel_sum_group_raw(lg, out=out, accum=acc))
            if not done:
                gb = ChannelSumG.apply(*lg)
    return gw, gb


def even_frames_ok(xs, w):
    """True when `conv_even_frames_group` can take these members: fp32 mode and the launch lands on the three-taps-per-round
    strip kernel (the only one that knows frame-strided outputs) — asked of the library's own plan query."""
    if len(xs) > MAX_GROUPS or w.dim() != 5 or not xs[0].is_cuda:
        return False
    cout, cin = w.shape[0], w.shape[1]
    k = tuple(w.shape[2:])
    geoms = [conv_geom(t.shape[0], cin, t.shape[2], t.shape[3], t.shape[4], cout, k[0], k[1], k[2]) for t in xs]
    mask = 0
    for g in geoms:
        mask |= g.mask
    ts = _tapset(geoms[0].T, mask)
    slot_of = {t: j for j, t in enumerate(ts.taps)}
    arr = _group_table(xs, [None] * len(xs), geoms, slot_of, None, 2)
    out = (C.c_int32 * 8)()
    if lib().t2v_conv_fwd_plan(arr, len(xs), cin, cout, FLAG_RELU_IN, out) != 0 or out[0] != 5 or out[7] != 1:
        return False
    # bf16-compute mode: its three-taps-per-round strip kernel knows the frame stride too (same strip conditions)
    return CONV_PRECISION == 'fp32' or bool(lib().t2v_conv_fwd_grouped_bf16_ok(arr, len(xs), cin, cout))


def _dgrad_even_frames_raw(gys, w5, masks, plan_only=False):
    """Data gradient of `ConvEvenFramesG` straight from dL/dy on the EVEN frames (gys[i]: [N,Cout,ceil(D/2),H,W]): output frame
    2e only sees the dz = 0 taps (9 of 27), frame 2e + 1 the dz = -1 / +1 taps reading dL/dy frames e / e + 1 (18) — half the
    MACs of the full-frame data gradient of the zero-stuffed tensor, and no zero-stuffing pass. Two strided-output launches
    (`t2v_conv_group.ydstride`) write every frame once; `masks` (the conv's input, for the fused ReLU adjoint) or None.
    Returns the list of gradients [N,Cin,D,H,W], or None when a launch would not land on the strip3 kernel (`plan_only`: True /
    None without launching anything)."""
    gys = [_c(g) for g in gys]
    Cout, Cin = w5.shape[0], w5.shape[1]
    if tuple(w5.shape[2:]) != (3, 3, 3) or len(gys) > MAX_GROUPS:
        return None
    bf16 = CONV_PRECISION == 'bf16'
    shapes = [tuple(m.shape) for m in masks]
    if any(sh[3] < 2 or sh[4] < 2 or g.shape[2] != (sh[2] + 1) // 2 for g, sh in zip(gys, shapes)):
        return None
    ts = _tapset(27, (1 << 27) - 1)
    slot_of = {t: j for j, t in enumerate(ts.taps)}
    outs = [torch.empty(sh, device=g.device, dtype=torch.float32) if not plan_only else g for g, sh in zip(gys, shapes)]
    launches = []
    for yoff, planes in ((0, ((1, 0),)), (1, ((0, 0), (2, 1)))):           # (kernel plane a of the mirrored weight, dL/dy frame offset)
        mem = [i for i, sh in enumerate(shapes) if (sh[2] + 1 - yoff) // 2 >= 1]
        if not mem:
            continue
        arr = (ConvGroup * len(mem))()
        for a_, i in zip(arr, mem):
            g, sh = gys[i], shapes[i]
            a_.x, a_.y, a_.mask = g.data_ptr(), outs[i].data_ptr(), masks[i].data_ptr()
            a_.N, a_.D, a_.H, a_.W = g.shape[0], g.shape[2], g.shape[3], g.shape[4]
            a_.dstride, a_.ydstride, a_.yoff, a_.Dy = 0, 2, yoff, sh[2]
            j = 0
            for pa, dzc in planes:
                for b in range(3):
                    for c in range(3):
                        a_.dz[j], a_.dy[j], a_.dx[j] = dzc, b - 1, c - 1
                        a_.widx[j] = slot_of[(pa * 3 + b) * 3 + c]
                        j += 1
            a_.ntaps = j
        plan = (C.c_int32 * 8)()
        if lib().t2v_conv_fwd_plan(arr, len(mem), Cout, Cin, FLAG_MASK_OUT, plan) != 0 or plan[0] != 5 or plan[7] != 1:
            return None
        if bf16 and not lib().t2v_conv_fwd_grouped_bf16_ok(arr, len(mem), Cout, Cin):
            return None
        launches.append((arr, len(mem)))
    if plan_only:
        return True
    if bf16:
        wpb = packed_weight_bf16(w5, ts, 1)
        for arr, n in launches:
            check(lib().t2v_conv_fwd_grouped_bf16(arr, n, Cout, Cin, _p(wpb), None, None, FLAG_MASK_OUT, _stream()), 't2v_conv_fwd_grouped_bf16')
        return outs
    wp = packed_weight(w5, ts, 1)
    for arr, n in launches:
        check(lib().t2v_conv_fwd_grouped(arr, n, Cout, Cin, _p(wp), None, None, FLAG_MASK_OUT, _stream()), 't2v_conv_fwd_grouped')
    return outs


class ConvDgradEvenMaskG(Function):
    """The recorded (gradient-penalty) form of `_dgrad_even_frames_raw`: gxs[i] = dgrad(zero-stuffed gys[i], w) * [xs[i] > 0] from
    the even-frame gradients. Its adjoints stay in even-frame form: d/d gys = the even frames of conv(masked ggx, w)
    (`ConvEvenFramesG`), d/d w = the even-frame weight gradient — conv / dgrad / wgrad stay a closed triple."""

    @staticmethod
    def forward(ctx, w, n, *gys_xs):
        gys, xs = gys_xs[:n], gys_xs[n:]
        ctx.save_for_backward(w, *gys_xs)
        ctx.set_materialize_grads(False)
        ctx.n = n
        out = _dgrad_even_frames_raw(gys, w, xs)
        if out is None:
            raise RuntimeError('ConvDgradEvenMaskG: launch plan changed between the check and the launch')
        return tuple(out)

    @staticmethod
    def backward(ctx, *ggxs):
        saved = ctx.saved_tensors
        n = ctx.n
        w, gys, xs = saved[0], saved[1:1 + n], saved[1 + n:]
        live = [i for i, g in enumerate(ggxs) if g is not None]
        d_w = None
        d_gys = [None] * n
        if live:
            hs = dict(zip(live, ReluMaskG.apply(len(live), *([ggxs[i] for i in live] + [xs[i] for i in live]))))
            if ctx.needs_input_grad[0] and _param_grads_enabled:
                lx, lg = [hs[i] for i in live], [gys[i] for i in live]
                done, d_w = _to_sink_w(w, lx, lg, False, even_frames=True)
                if not done:          # no sink (or a graph is being recorded): through the zero-stuffed gradient
                    cfg = ((1, 1, 1), (2, 1, 1), (0, 0, 0))
                    full = AvgPool3dBwdG.apply(tuple(cfg for _ in live), tuple(tuple(xs[i].shape[2:]) for i in live), *lg)
                    d_w = ConvWgradG.apply(tuple(w.shape), False, len(live), *(lx + list(full)))
            need = [i for i in live if ctx.needs_input_grad[2 + i]]
            if need:
                for i, r in zip(need, ConvEvenFramesG.apply(w, None, False, *[hs[i] for i in need])):
                    d_gys[i] = r
        return (d_w, None) + tuple(d_gys) + (None,) * n


class ConvEvenFramesG(Function):
    """ys[i] = conv(relu?(xs[i]), w)[:, :, ::2] (+ b): the stem's second convolution feeds `AvgPool3d((1,2,2), stride 2)`
    (resnet3d.py:12-19), which keeps the EVEN frames only — the odd output frames are not computed (half the forward GEMM).
    The adjoint is the plain convolution's, applied to the gradient scattered back to the even frames (zeros elsewhere), i.e.
    exactly what the pooling adjoint used to hand over; composed of the same differentiable Functions as `ConvG.backward`."""

    @staticmethod
    def forward(ctx, w, b, relu_in, *xs):
        ctx.save_for_backward(w, *xs)
        ctx.set_materialize_grads(False)
        ctx.has_bias, ctx.relu_in, ctx.bias = b is not None, relu_in, b
        return tuple(conv_group_raw(xs, w, b, relu_in, 0, even_frames=True))

    @staticmethod
    def backward(ctx, *gys):
        saved = ctx.saved_tensors
        w, xs = saved[0], saved[1:]
        live = [i for i, g in enumerate(gys) if g is not None]
        gxs = [None] * len(xs)
        gw = gb = None
        if not live:
            return (None, None, None) + tuple(gxs)
        need = [i for i in live if ctx.needs_input_grad[3 + i]]
        fast = None
        if need and ctx.relu_in and not torch.is_grad_enabled():
            # no graph is being recorded: the data gradient straight from the even-frame gradients (two strided-output launches)
            fast = _dgrad_even_frames_raw([gys[i] for i in need], w, [xs[i] for i in need])
            if fast is not None:
                for i, r in zip(need, fast):
                    gxs[i] = r
        gfull = [None] * len(xs)

        def full_frames():
            # dL/dy on all frames: the even frames carry the gradient, the odd ones zeros (the adjoint of y[:, :, ::2])
            if all(gfull[i] is None for i in live):
                cfg = ((1, 1, 1), (2, 1, 1), (0, 0, 0))
                full = AvgPool3dBwdG.apply(tuple(cfg for _ in live), tuple(tuple(xs[i].shape[2:]) for i in live), *[gys[i] for i in live])
                for i, g in zip(live, full):
                    gfull[i] = g
            return gfull
        if need and fast is None and ctx.relu_in and torch.is_grad_enabled() and \
                _dgrad_even_frames_raw([gys[i] for i in need], w, [xs[i] for i in need], plan_only=True):
            # a graph is being recorded (gradient penalty): the same two launches as a differentiable Function
            res = ConvDgradEvenMaskG.apply(w, len(need), *([gys[i] for i in need] + [xs[i] for i in need]))
            for i, r in zip(need, res):
                gxs[i] = r
            fast = res
        if need and fast is None:
            full_frames()
            if ctx.relu_in:
                res = ConvDgradMaskG.apply(w, len(need), *([gfull[i] for i in need] + [xs[i] for i in need]))
            else:
                res = ConvDgradG.apply(w, *[gfull[i] for i in need])
            for i, r in zip(need, res):
                gxs[i] = r
        if _param_grads_enabled:
            need_w, need_b = ctx.needs_input_grad[0], ctx.has_bias and ctx.needs_input_grad[1]
            lx, lg = [xs[i] for i in live], [gys[i] for i in live]
            done = False
            if need_w and not torch.is_grad_enabled() and w.shape[1] >= 64 and w.shape[4] == 3 and any(t.shape[4] > 1 for t in lx):
                # gradient sink armed: the weight-gradient kernel reads dL/dy on the even frames as it is (half the voxels)
                if need_b:
                    done, gw, gb = _to_sink_wb(w, ctx.bias, lx, lg, ctx.relu_in, even_frames=True)
                else:
                    done, gw = _to_sink_w(w, lx, lg, ctx.relu_in, even_frames=True)
            if not done:
                full_frames()
                gw, gb = _group_param_grads(w, ctx.bias if ctx.has_bias else None, ctx.relu_in, lx, [gfull[i] for i in live], need_w, need_b)
        return (gw, gb, None) + tuple(gxs)


def conv_even_frames_group(xs, w, b=None, relu_in=False):
    return list(ConvEvenFramesG.apply(w, b, relu_in, *xs))


class ConvMultiG(Function):
    """Several convolutions of the SAME tensors (a block's main and skip convolution; the non-local block's theta / phi / g):
    outputs of spec k are ys[k*n:(k+1)*n]. Forward = one grouped launch per spec, as separate ConvG calls would be; the
    point is the backward: the data gradients of all specs are accumulated by the kernels into ONE buffer per member
    (T2V_CONV_ACCUM), so autograd never has to add the contributions of the consumers of a tensor. While autograd is
    recording (the gradient penalty's sweep) the backward is composed from the differentiable Functions instead.
    args: relus (tuple of bool per spec), nspec, then w_0, b_0, ..., w_{k-1}, b_{k-1}, then the n tensors."""

    @staticmethod
    def forward(ctx, relus, nspec, *rest):
        wb, xs = rest[:2 * nspec], rest[2 * nspec:]
        ws, bs = wb[0::2], wb[1::2]
        ctx.save_for_backward(*ws, *xs)
        ctx.set_materialize_grads(False)
        ctx.cfg = (relus, nspec, len(xs))
        ctx.biases = bs
        out = []
        for w, b, r in zip(ws, bs, relus):
            out += conv_group_raw(xs, w, b, r, 0)
        return tuple(out)

    @staticmethod
    def backward(ctx, *gys):
        relus, nspec, n = ctx.cfg
        saved = ctx.saved_tensors
        ws, xs = saved[:nspec], saved[nspec:]
        bs = ctx.biases
        g_of = [gys[k * n:(k + 1) * n] for k in range(nspec)]
        need_x = [ctx.needs_input_grad[2 + 2 * nspec + i] for i in range(n)]
        gxs = [None] * n
        order = sorted(range(nspec), key=lambda k: 0 if relus[k] else 1)           # masked data gradients write first
        fused = not torch.is_grad_enabled()
        for k in order:
            mem = [i for i in range(n) if g_of[k][i] is not None and need_x[i]]
            if not mem:
                continue
            fresh = [i for i in mem if gxs[i] is None]
            again = [i for i in mem if gxs[i] is not None]
            if fresh:
                if relus[k]:
                    res = ConvDgradMaskG.apply(ws[k], len(fresh), *([g_of[k][i] for i in fresh] + [xs[i] for i in fresh]))
                else:
                    res = ConvDgradG.apply(ws[k], *[g_of[k][i] for i in fresh])
                for i, r in zip(fresh, res):
                    gxs[i] = r
            if again:
                if fused and not relus[k]:
                    conv_group_raw([g_of[k][i] for i in again], ws[k], None, False, 1, outs=[gxs[i] for i in again], accum=True)
                else:
                    if relus[k]:
                        res = ConvDgradMaskG.apply(ws[k], len(again), *([g_of[k][i] for i in again] + [xs[i] for i in again]))
                    else:
                        res = ConvDgradG.apply(ws[k], *[g_of[k][i] for i in again])
                    # (recorded backward: one grouped add for all members instead of one launch per member)
                    for i, t in zip(again, AddG.apply(*([gxs[i] for i in again] + list(res)))):
                        gxs[i] = t
        grads = []
        for k in range(nspec):
            gw = gb = None
            live = [i for i in range(n) if g_of[k][i] is not None]
            if live and _param_grads_enabled:
                gw, gb = _group_param_grads(ws[k], bs[k], relus[k], [xs[i] for i in live], [g_of[k][i] for i in live],
                                            ctx.needs_input_grad[2 + 2 * k], bs[k] is not None and ctx.needs_input_grad[3 + 2 * k])
            grads += [gw, gb]
        return (None, None) + tuple(grads) + tuple(gxs)


def conv_multi_group(xs, specs):
    """specs: [(w, b, relu_in), ...] applied to the same tensors; returns one list of outputs per spec."""
    n = len(xs)
    flat = []
    for w, b, _ in specs:
        flat += [w, b]
    out = ConvMultiG.apply(tuple(bool(r) for _, _, r in specs), len(specs), *(flat + list(xs)))
    return [list(out[k * n:(k + 1) * n]) for k in range(len(specs))]


def conv_out_like(x, w):
    return torch.empty((x.shape[0], w.shape[0]) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)


class ConvDgradG(Function):
    @staticmethod
    def forward(ctx, w, *gys):
        ctx.save_for_backward(w, *gys)
        ctx.set_materialize_grads(False)
        return tuple(conv_group_raw(gys, w, None, False, 1))

    @staticmethod
    def backward(ctx, *ggxs):
        saved = ctx.saved_tensors
        w, gys = saved[0], saved[1:]
        live = [i for i, g in enumerate(ggxs) if g is not None]
        d_w = None
        d_gys = [None] * len(gys)
        if not live:
            return (None,) + tuple(d_gys)
        if ctx.needs_input_grad[0] and _param_grads_enabled:
            lx, lg = [ggxs[i] for i in live], [gys[i] for i in live]
            done, d_w = _to_sink_w(w, lx, lg, False)
            if not done:
                d_w = ConvWgradG.apply(tuple(w.shape), False, len(live), *(lx + lg))
        need = [i for i in live if ctx.needs_input_grad[1 + i]]
        if need:
            res = ConvG.apply(w, None, False, *[ggxs[i] for i in need])
            for i, r in zip(need, res):
                d_gys[i] = r
        return (d_w,) + tuple(d_gys)


class ConvDgradMaskG(Function):
    """gxs[i] = dgrad(gys[i], w) * [xs[i] > 0]: the data gradient of a ReLU -> conv pair with the ReLU adjoint applied in
    the epilogue (T2V_CONV_MASK_OUT) — one launch, no unmasked intermediate. Its own adjoints (the gradient penalty
    differentiates through it) are composed from the unfused differentiable ops."""

    @staticmethod
    def forward(ctx, w, n, *gys_xs):
        gys, xs = gys_xs[:n], gys_xs[n:]
        ctx.save_for_backward(w, *gys_xs)
        ctx.set_materialize_grads(False)
        ctx.n = n
        return tuple(conv_group_raw(gys, w, None, False, 1, masks=xs))

    @staticmethod
    def backward(ctx, *ggxs):
        saved = ctx.saved_tensors
        n = ctx.n
        w, gys, xs = saved[0], saved[1:1 + n], saved[1 + n:]
        live = [i for i, g in enumerate(ggxs) if g is not None]
        d_w = None
        d_gys = [None] * n
        if live:
            if len(live) > 1:
                hs = dict(zip(live, ReluMaskG.apply(len(live), *([ggxs[i] for i in live] + [xs[i] for i in live]))))
            else:
                hs = {i: ReluMask.apply(ggxs[i], xs[i]) for i in live}
            if ctx.needs_input_grad[0] and _param_grads_enabled:
                lx, lg = [hs[i] for i in live], [gys[i] for i in live]
                done, d_w = _to_sink_w(w, lx, lg, False)
                if not done:
                    d_w = ConvWgradG.apply(tuple(w.shape), False, len(live), *(lx + lg))
            need = [i for i in live if ctx.needs_input_grad[2 + i]]
            if need:
                res = ConvG.apply(w, None, False, *[hs[i] for i in need])
                for i, r in zip(need, res):
                    d_gys[i] = r
        return (d_w, None) + tuple(d_gys) + (None,) * n


class ConvWgradG(Function):
    """gw = sum over the group of wgrad(relu?(x_i), gy_i).""" 

    @staticmethod
    def forward(ctx, wshape, relu_in, n, *xs_gys):
        xs, gys = xs_gys[:n], xs_gys[n:]
        ctx.save_for_backward(*xs_gys)
        ctx.cfg = (wshape, relu_in, n)
        return conv_group_wgrad_raw(xs, gys, wshape, relu_in)

    @staticmethod
    def backward(ctx, ggw):
        wshape, relu_in, n = ctx.cfg
        saved = ctx.saved_tensors
        xs, gys = saved[:n], saved[n:]
        d_xs = [None] * n
        d_gys = [None] * n
        need_x = [i for i in range(n) if ctx.needs_input_grad[3 + i]]
        if need_x:
            res = ConvDgradG.apply(ggw, *[gys[i] for i in need_x])
            for i, r in zip(need_x, res):
                d_xs[i] = ReluMask.apply(r, xs[i]) if relu_in else r
        need_g = [i for i in range(n) if ctx.needs_input_grad[3 + n + i]]
        if need_g:
            res = ConvG.apply(ggw, None, relu_in, *[xs[i] for i in need_g])
            for i, r in zip(need_g, res):
                d_gys[i] = r
        return (None, None, None) + tuple(d_xs) + tuple(d_gys)


def channel_sum_group_raw(gys, out=None, accum=False):
    gys = [_c(g) for g in gys]
    Cc = gys[0].shape[1]
    arr = (ConvGroup * len(gys))()
    for a, g in zip(arr, gys):
        a.x, a.N = g.data_ptr(), g.shape[0]
        a.D, a.H, a.W = g.shape[2], g.shape[3], g.shape[4]
    if out is None:
        out = torch.empty((Cc,), device=gys[0].device, dtype=torch.float32)
    nws = int(lib().t2v_channel_sum_grouped_ws_floats(arr, len(gys), Cc))
    ws = torch.empty((nws,), device=out.device, dtype=torch.float32) if nws > 0 else None
    check(lib().t2v_channel_sum_grouped(arr, len(gys), Cc, _p(out), _p(ws), 1 if accum else 0, _stream()), 't2v_channel_sum_grouped')
    return out


class ChannelSumG(Function):
    """bias gradient of a grouped convolution: sum over every member, one launch."""

    @staticmethod
    def forward(ctx, *gys):
        return channel_sum_group_raw(gys)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raise RuntimeError('bias gradients are first-order only on this path')


def conv_group(xs, w, b=None, relu_in=False):
    """[conv(relu?(x), w) + b for x in xs] with ONE kernel launch (5-D tensors sharing w)."""
    return list(ConvG.apply(w, b, relu_in, *xs))


# ------------------------------------------------------------------------------------------------
# Undefined gradients stay undefined. A grouped convolution returns None for the members nobody
# differentiates; by default autograd would turn that None into a zeros tensor at the next per-member node
# and the whole (useless) backward of that member would run on zeros. Every single-output Function of the
# discriminator path therefore declares `set_materialize_grads(False)` and passes None straight through.
# ------------------------------------------------------------------------------------------------

def _pass_none_through(cls):
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx.set_materialize_grads(False)
        ctx._t2v_n_in = len(args)
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return (None,) * ctx._t2v_n_in
        return bwd(ctx, *grads)
    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


# the pooled second convolution of the discriminator blocks (box-sum + stride-2 GEMMs): functional_pool.py
from .functional_pool import (pool_conv_ok, pool_conv_group, pool_tmode, PoolConvG, PoolConvDgradG, PoolConvWgradG, boxsum_raw, unbox_raw,     # noqa: E402,F401
                              pool_fwd_raw, pool_dgrad_raw, pool_wgrad_raw, up_conv, up_conv_ok, UpConvFn)


def add_group(as_, bs):
    """[a + b for a, b in zip(as_, bs)] in one launch (`AddG`)."""
    return list(AddG.apply(*(list(as_) + list(bs))))


for _cls in (Conv, ConvDgrad, ConvWgrad, ReluConv, ReluConvWgrad, Relu, ReluMask, Add, AvgPool3d, AddAvgPool3d, AvgPool3dBwd, MaxPool2x2,
             MaxScatter, MaxGather, RowSum, RowBcast, Bmm, Softmax, SoftmaxBwd, Dot, ScaleDev, CatFeatures, SliceCols,
             EmbedCols, CatBatch, ConvWgradG):
    _pass_none_through(_cls)
