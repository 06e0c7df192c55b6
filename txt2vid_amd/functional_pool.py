"""Pooled convolution: the second 3^3 convolution of a discriminator block TOGETHER with the average pooling that is its only
consumer (txt2vid/models/resnet3d.py:12-19 stem: `conv -> AvgPool3d((1,2,2), 2)`; txt2vid/models/layers.py:219-243 DownBlock:
`conv -> DownSample`). Box filter and convolution commute, so

    pool(conv3(relu(r))) = stride-2 conv3 of r~,   r~[p] = scale * sum_{delta in {0,1}^k} relu(r)[p - 1 + delta]

— 27 taps over the POOLED voxels instead of 27 taps over all of them: 4x fewer MACs behind the stem's (1,2,2) pooling (which
also keeps the even frames only), 8x behind a (2,2,2) DownSample; identical values up to fp32 summation order. The three GEMMs
(forward, data gradient, weight gradient: `t2v_pool_conv_*`, conv.hip) and the two streaming passes (`t2v_pool_boxsum`,
`t2v_pool_unbox`, pointwise.hip) form a triple closed under differentiation like conv / dgrad / wgrad, so the gradient
penalty's recorded sweep and its second backward stay on these kernels. Part of the autograd surface `txt2vid_amd.functional`
(re-exported there)."""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import lib, check, ConvGroup, PoolBoxJob, WgradSrc, MAX_GROUPS
from ._ops import _stream, _p, _c

_DISABLED = os.environ.get('T2V_NO_POOLCONV') is not None        # developer A/B switch: the un-pooled path everywhere
_NO_UPCONV = os.environ.get('T2V_NO_UPCONV') is not None         # developer A/B switch: UpBlocks up-sample, then convolve


def _TF():
    from . import functional
    return functional


def pool_tmode(shape, stem):
    """How the time axis of a [N,C,D,H,W] member is treated, or None when the pooled form does not apply to it:
    0 no time axis (D == 1), 1 box + stride 2 (DownSample), 2 stride 2 without a box (the stem's pooling keeps the even frames)."""
    D, H, W = shape[2], shape[3], shape[4]
    if H < 2 or W < 2 or H % 2 or W % 2:
        return None
    if D == 1:
        return 0
    if D % 2:
        return None
    return 2 if stem else 1


def pool_scale(tmode):
    return 0.125 if tmode == 1 else 0.25


def pool_conv_ok(xs, w, stem):
    """True when `pool_conv_group` takes these members: 3x3x3 kernel, channels a multiple of 32, even H / W >= 2, D == 1 or even,
    fp32 mode. (bf16-compute mode keeps the un-pooled layers: measured, the pooled fp32 GEMMs + their two streaming passes lose to
    the un-pooled bf16 GEMMs — 9.66 vs 9.19 ms per iteration with every layer pooled, 9.43 with only the DownBlocks' 8x layers, 9.39
    with only the UpBlocks' transposed form; a bf16 form of the pooled kernels is the open item there.)"""
    TF = _TF()
    if _DISABLED or TF.CONV_PRECISION != 'fp32' or not (1 <= len(xs) <= MAX_GROUPS) or w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        return False
    if w.shape[0] % 32 or w.shape[1] % 32 or not xs[0].is_cuda:      # (both: the data gradient contracts over Cout)
        return False
    return all(x.dim() == 5 and pool_tmode(x.shape, stem) is not None for x in xs)


def pooled_shape(shape, tmode):
    N, _, D, H, W = shape
    return (N, D // 2 if tmode else 1, H // 2, W // 2)


# ------------------------------------------------------------------------------------------------
# raw launches
# ------------------------------------------------------------------------------------------------

def _box_jobs(ins, masks, outs, shapes, tmodes, relu, scale=None, bias=None):
    """`relu`: boxsum: clamp the input at 0; unbox: the number of plane sets to add up."""
    arr = (PoolBoxJob * len(ins))()
    for a, t, sh, tm, i in zip(arr, ins, shapes, tmodes, range(len(ins))):
        a.in_, a.out = t.data_ptr(), outs[i].data_ptr()
        a.mask = masks[i].data_ptr() if masks is not None else None
        a.bias = bias.data_ptr() if bias is not None else None
        a.NC, a.D, a.H, a.W, a.C = sh[0] * sh[1], sh[2], sh[3], sh[4], sh[1]
        a.tmode, a.relu, a.scale = tm, int(relu), pool_scale(tm) if scale is None else scale
    return arr


def boxsum_raw(ins, tmodes, relu, masks=None, scale=None):
    """r~ of every member (padded grid [N,C,Dp,H+1,W+2]); `masks`: the inputs are cotangents, multiplied by [mask > 0] first.
    `scale`: instead of 1 / window volume (the up-sampling form passes 1)."""
    ins = [_c(t) for t in ins]
    masks = [_c(m) for m in masks] if masks is not None else None
    shapes = [tuple(t.shape) for t in ins]
    outs = [torch.empty((sh[0], sh[1], sh[2] + 1 if tm else 1, sh[3] + 1, sh[4] + 2), device=t.device, dtype=torch.float32)
            for t, sh, tm in zip(ins, shapes, tmodes)]
    check(lib().t2v_pool_boxsum(_box_jobs(ins, masks, outs, shapes, tmodes, relu, scale), len(ins), _stream()), 't2v_pool_boxsum')
    return outs


def unbox_raw(planes, shapes, tmodes, masks=None, scale=None, bias=None):
    """The adjoint of `boxsum_raw` applied to the 8 class planes of each member; `masks`: result zeroed where mask <= 0;
    `bias`: a per-channel constant added on top (the up-sampling form's convolution bias)."""
    masks = [_c(m) for m in masks] if masks is not None else None
    outs = [torch.empty(tuple(sh), device=p.device, dtype=torch.float32) for p, sh in zip(planes, shapes)]
    sets = planes[0].shape[0] if planes[0].dim() == 7 else 1
    check(lib().t2v_pool_unbox(_box_jobs(planes, masks, outs, shapes, tmodes, sets, scale, bias), len(planes), _stream()), 't2v_pool_unbox')
    return outs


def _tap_union(tmodes, kD=3):
    """The taps to pack: all 27, or the dz = 0 plane when no member has a time axis; the 9 taps of a [.,.,1,3,3] weight (kD = 1)."""
    TF = _TF()
    if kD == 1:
        if any(tmodes):
            raise ValueError('a 2-D kernel cannot serve members with a time axis')
        return TF._tapset(9, 0x1ff)
    mask = (1 << 27) - 1 if any(tmodes) else 0x1ff << 9
    return TF._tapset(27, mask)


def _tap_index(ts, dz, dy, dx):
    return (dy + 1) * 3 + dx + 1 if ts.T == 9 else ((dz + 1) * 3 + dy + 1) * 3 + dx + 1


def _fwd_table(xs, ys, shapes, tmodes, ts):
    slot_of = {t: j for j, t in enumerate(ts.taps)}
    arr = (ConvGroup * len(xs))()
    for a, x, y, sh, tm in zip(arr, xs, ys, shapes, tmodes):
        a.x, a.y, a.mask = x.data_ptr(), (y.data_ptr() if y is not None else 0), None
        a.N, a.D, a.H, a.W = sh[0], sh[2], sh[3], sh[4]
        a.dstride = tm
        j = 0
        for dz in ((-1, 0, 1) if tm else (0,)):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    a.dz[j], a.dy[j], a.dx[j] = dz, dy, dx
                    a.widx[j] = slot_of[_tap_index(ts, dz, dy, dx)]
                    j += 1
        a.ntaps = j
    return arr


def pool_fwd_raw(rts, shapes, tmodes, w5, bias=None, transpose=False):
    """zs[i] = stride-2 conv3 of the padded box sums rts[i] (+ bias): [N, Cout, D', H', W']. `shapes`: the full-resolution
    [N,C,D,H,W] of each member. `transpose`: contract over the weight's OUTPUT channels with mirrored taps (mode-1 packing) — the
    data gradient of the up-sampling form."""
    TF = _TF()
    w5 = _c(w5)
    cout, cin = (w5.shape[1], w5.shape[0]) if transpose else (w5.shape[0], w5.shape[1])
    ts = _tap_union(tmodes, w5.shape[2])
    zs = []
    for sh, tm, t in zip(shapes, tmodes, rts):
        n, d, h, w_ = pooled_shape(sh, tm)
        zs.append(torch.empty((n, cout, d, h, w_), device=t.device, dtype=torch.float32))
    arr = _fwd_table(rts, zs, shapes, tmodes, ts)
    nws = int(lib().t2v_pool_conv_fwd_ws_floats(arr, len(rts), cin, cout))
    if nws < 0:
        raise RuntimeError('bad pooled-convolution geometry')
    ws = torch.empty((nws,), device=rts[0].device, dtype=torch.float32) if nws > 0 else None
    wp = TF.packed_weight(w5, ts, 1 if transpose else 0)
    check(lib().t2v_pool_conv_fwd(arr, len(rts), cin, cout, _p(wp), _p(bias), _p(ws), TF.FLAG_BIAS if bias is not None else 0, _stream()),
          't2v_pool_conv_fwd')
    return zs


def pool_dgrad_raw(gzs, shapes, tmodes, w5, transpose=False):
    """The parity-class planes [S, 8, N, Cin, Dq, H/2+1, W/2+1] of the data gradient on the padded grid, from dL/dy (pooled shape);
    S = the launch's k-split count (`unbox_raw` adds the S sets up).
    `transpose`: contract over the weight's INPUT channels instead (mode-0 packing): the class planes of `conv3(zero-stuffed x)`,
    i.e. the forward of the up-sampling form (then `gzs` are the small maps and the planes have Cout channels)."""
    TF = _TF()
    gzs = [_c(g) for g in gzs]
    w5 = _c(w5)
    cout, cin = (w5.shape[1], w5.shape[0]) if transpose else (w5.shape[0], w5.shape[1])          # (K of the GEMM, channels of the planes)
    ts = _tap_union(tmodes, w5.shape[2])
    slot_of = {t: j for j, t in enumerate(ts.taps)}
    arr = (ConvGroup * len(gzs))()
    for a, g, sh, tm in zip(arr, gzs, shapes, tmodes):
        if tuple(g.shape) != (sh[0], cout) + pooled_shape(sh, tm)[1:]:
            raise ValueError('pooled gradient of shape %s for a member of shape %s' % (tuple(g.shape), tuple(sh)))
        a.x, a.y, a.mask = g.data_ptr(), None, None
        a.N, a.D, a.H, a.W = sh[0], sh[2], sh[3], sh[4]
        a.dstride, a.ntaps = tm, 27
        for f in range(27):                                       # the kernel numbers its taps f = ((dz+1)*3 + dy+1)*3 + dx+1
            # mode-1 packing stores forward tap f in the slot of its mirror; in the transposed (mode-0) role the class structure
            # itself asks for the mirrored tap: the same index either way
            a.widx[f] = slot_of.get(26 - f, -1) if ts.T == 27 else (slot_of.get(8 - (f - 9), -1) if 9 <= f < 18 else -1)
    S = int(lib().t2v_pool_conv_dgrad_splits(arr, len(gzs), cout, cin))           # k-split: S sets of planes, summed by the unbox pass
    if S < 1:
        raise RuntimeError('bad pooled data-gradient geometry')
    planes = [torch.empty((S, 8, sh[0], cin, sh[2] // 2 + 1 if tm else 1, sh[3] // 2 + 1, sh[4] // 2 + 1), device=g.device, dtype=torch.float32)
              for g, sh, tm in zip(gzs, shapes, tmodes)]
    for a, pl in zip(arr, planes):
        a.y = pl.data_ptr()
    wp = TF.packed_weight(w5, ts, 0 if transpose else 1)
    check(lib().t2v_pool_conv_dgrad(arr, len(gzs), cout, cin, _p(wp), _stream()), 't2v_pool_conv_dgrad')
    return planes


def _wgrad_table(rts, gzs, shapes, tmodes):
    arr = (ConvGroup * len(rts))()
    for a, x, g, sh, tm in zip(arr, rts, gzs, shapes, tmodes):
        a.x, a.y, a.mask = x.data_ptr(), g.data_ptr(), None
        a.N, a.D, a.H, a.W = sh[0], sh[2], sh[3], sh[4]
        a.dstride, a.ntaps = tm, 27 if tm else 9
    return arr


def pool_wgrad_raw(rts, gzs, shapes, tmodes, wshape, out=None, accum=False, dbias=None, accum_bias=False):
    """dW (and the bias gradient when `dbias` is given) summed over the members, reduce pass included. wshape: [Cout,Cin,kD,3,3]."""
    TF = _TF()
    gzs = [_c(g) for g in gzs]
    cout, cin, kD = wshape[0], wshape[1], wshape[2]
    arr = _wgrad_table(rts, gzs, shapes, tmodes)
    n = int(lib().t2v_pool_conv_wgrad_slab_floats(arr, len(rts), cin, cout, kD, 1 if dbias is not None else 0))
    if n <= 0:
        raise RuntimeError('bad pooled weight-gradient geometry')
    slab = torch.empty((n,), device=rts[0].device, dtype=torch.float32)
    dw = out if out is not None else torch.empty(tuple(wshape), device=rts[0].device, dtype=torch.float32)
    flags = (TF.FLAG_ACCUM if accum else 0) | (TF.FLAG_ACCUM_BIAS if accum_bias else 0)
    check(lib().t2v_pool_conv_wgrad(arr, len(rts), cin, cout, kD, _p(dw), _p(dbias), _p(slab), flags, _stream()), 't2v_pool_conv_wgrad')
    return dw


def _pool_wgrad_to_sink(w, b, rts, gzs, shapes, tmodes, need_b):
    """Weight (and bias) gradient into the gradient sink's slots, deferred like `functional.conv_group_wgrad_partial` when the sink
    batches its reductions. Returns (handled, gw, gb)."""
    TF = _TF()
    sink = TF._grad_sink
    if sink is None or torch.is_grad_enabled():
        return False, None, None
    wbase = w._base if w._base is not None else w
    if id(wbase) not in sink.slots or (need_b and id(b) not in sink.slots):
        return False, None, None
    gzs = [_c(g) for g in gzs]
    cout, cin = w.shape[0], w.shape[1]
    wflat, wacc = sink.take(wbase, sink.defer)
    bflat, bacc = sink.take(b, sink.defer) if need_b else (None, False)
    if sink.defer:
        arr = _wgrad_table(rts, gzs, shapes, tmodes)
        n = int(lib().t2v_pool_conv_wgrad_slab_floats(arr, len(rts), cin, cout, 3, 1 if need_b else 0))
        if n <= 0:
            raise RuntimeError('bad pooled weight-gradient geometry')
        slab = sink.alloc_slab(n, rts[0].device)
        src = WgradSrc()
        check(lib().t2v_pool_conv_wgrad_partial(arr, len(rts), cin, cout, 3, _p(slab), 1 if need_b else 0, 0, C.byref(src),
                                                TF._side.fork(rts, gzs)), 't2v_pool_conv_wgrad_partial')
        sink.add_partial(wbase, wflat.view(w.shape), wacc, b if need_b else None, bflat, bacc, src, slab, 27, cout, cout * cin)
    else:
        pool_wgrad_raw(rts, gzs, shapes, tmodes, tuple(w.shape), out=wflat.view(w.shape), accum=wacc, dbias=bflat, accum_bias=bacc)
    return True, (None if wacc else wflat.view(w.shape)), ((None if bacc else bflat.view(b.shape)) if need_b else None)


# ------------------------------------------------------------------------------------------------
# autograd surface
# ------------------------------------------------------------------------------------------------

class PoolConvG(Function):
    """ys[i] = pool_i(conv3(relu?(xs[i]), w)) + b, computed as the stride-2 convolution of the box-summed activation.
    args: w, b, relu_in, tmodes (tuple), then the members. Members that receive no gradient cost nothing in the backward."""

    @staticmethod
    def forward(ctx, w, b, relu_in, tmodes, *xs):
        xs = [_c(x) for x in xs]
        shapes = [tuple(x.shape) for x in xs]
        rts = boxsum_raw(xs, tmodes, relu_in)
        ctx.save_for_backward(w, *xs, *rts)
        ctx.set_materialize_grads(False)
        ctx.cfg = (relu_in, tuple(tmodes), shapes, b)
        return tuple(pool_fwd_raw(rts, shapes, tmodes, w, b))

    @staticmethod
    def backward(ctx, *gzs):
        TF = _TF()
        relu_in, tmodes, shapes, b = ctx.cfg
        saved = ctx.saved_tensors
        n = len(shapes)
        w, xs, rts = saved[0], saved[1:1 + n], saved[1 + n:]
        live = [i for i, g in enumerate(gzs) if g is not None]
        gxs = [None] * n
        gw = gb = None
        if not live:
            return (None, None, None, None) + tuple(gxs)
        need = [i for i in live if ctx.needs_input_grad[4 + i]]
        if need:
            res = PoolConvDgradG.apply(w, relu_in, tuple(tmodes[i] for i in need), len(need), *([gzs[i] for i in need] + [xs[i] for i in need]))
            for i, r in zip(need, res):
                gxs[i] = r
        if TF._param_grads_enabled:
            need_w, need_b = ctx.needs_input_grad[0], b is not None and ctx.needs_input_grad[1]
            if need_w:
                lr, lg = [rts[i] for i in live], [gzs[i] for i in live]
                lsh, ltm = [shapes[i] for i in live], [tmodes[i] for i in live]
                done, gw, gb = _pool_wgrad_to_sink(w, b, lr, lg, lsh, ltm, need_b)
                if not done:
                    gw = PoolConvWgradG.apply(tuple(w.shape), tuple(ltm), tuple(lsh), len(live), *(lr + lg))
                    if need_b:
                        gb = TF.ChannelSumG.apply(*lg)
            elif need_b:
                gb = TF.ChannelSumG.apply(*[gzs[i] for i in live])
        return (gw, gb, None, None) + tuple(gxs)


class PoolConvDgradG(Function):
    """gxs[i] = [xs[i] > 0]? * boxsum^T(conv^T(subsample^T(gzs[i]))): the data gradient of `PoolConvG` from the POOLED gradient
    (no zero-stuffed or un-pooled intermediate). Its adjoints: d/d gz = the pooled convolution of the masked cotangent, d/d w =
    the pooled weight gradient of (box-summed masked cotangent, gz) — the triple is closed under differentiation.
    args: w, masked, tmodes, n, gz_0.., x_0.. (the x are the ReLU masks; unused when `masked` is False)."""

    @staticmethod
    def forward(ctx, w, masked, tmodes, n, *gzs_xs):
        gzs, xs = gzs_xs[:n], gzs_xs[n:]
        shapes = [tuple(x.shape) for x in xs]
        ctx.save_for_backward(w, *gzs_xs)
        ctx.set_materialize_grads(False)
        ctx.cfg = (masked, tuple(tmodes), shapes, n)
        planes = pool_dgrad_raw(gzs, shapes, tmodes, w)
        return tuple(unbox_raw(planes, shapes, tmodes, masks=xs if masked else None))

    @staticmethod
    def backward(ctx, *ggxs):
        TF = _TF()
        masked, tmodes, shapes, n = ctx.cfg
        saved = ctx.saved_tensors
        w, gzs, xs = saved[0], saved[1:1 + n], saved[1 + n:]
        live = [i for i, g in enumerate(ggxs) if g is not None]
        d_w = None
        d_gzs = [None] * n
        if not live:
            return (None, None, None, None) + tuple(d_gzs) + (None,) * n
        ltm, lsh = [tmodes[i] for i in live], [shapes[i] for i in live]
        need = [i for i in live if ctx.needs_input_grad[4 + i]]
        want_w = ctx.needs_input_grad[0] and TF._param_grads_enabled
        if not torch.is_grad_enabled():
            # (the usual case: the second backward of the gradient penalty is not itself recorded) one box-sum pass with the
            # mask fused serves both adjoints
            rts = boxsum_raw([ggxs[i] for i in live], ltm, False, masks=[xs[i] for i in live] if masked else None)
            if need:
                pos = [live.index(i) for i in need]
                res = pool_fwd_raw([rts[k] for k in pos], [lsh[k] for k in pos], [ltm[k] for k in pos], w, None)
                for i, r in zip(need, res):
                    d_gzs[i] = r
            if want_w:
                done, d_w, _ = _pool_wgrad_to_sink(w, None, rts, [gzs[i] for i in live], lsh, ltm, False)
                if not done:
                    d_w = pool_wgrad_raw(rts, [gzs[i] for i in live], lsh, ltm, tuple(w.shape))
        else:
            hs = list(TF.ReluMaskG.apply(len(live), *([ggxs[i] for i in live] + [xs[i] for i in live]))) if masked else [ggxs[i] for i in live]
            if need:
                pos = [live.index(i) for i in need]
                res = PoolConvG.apply(w, None, False, tuple(ltm[k] for k in pos), *[hs[k] for k in pos])
                for i, r in zip(need, res):
                    d_gzs[i] = r
            if want_w:
                rts = boxsum_raw([h.detach() for h in hs], ltm, False)
                d_w = PoolConvWgradG.apply(tuple(w.shape), tuple(ltm), tuple(lsh), len(live), *(rts + [gzs[i] for i in live]))
        return (d_w, None, None, None) + tuple(d_gzs) + (None,) * n


class PoolConvWgradG(Function):
    """gw = sum over the members of the pooled weight gradient (immediate reduce): the path without a gradient sink."""

    @staticmethod
    def forward(ctx, wshape, tmodes, shapes, n, *rts_gzs):
        rts, gzs = rts_gzs[:n], rts_gzs[n:]
        return pool_wgrad_raw(list(rts), list(gzs), list(shapes), list(tmodes), wshape)

    @staticmethod
    @once_differentiable
    def backward(ctx, ggw):
        raise RuntimeError('the pooled weight gradient is differentiated at most once on this path')


def pool_conv_group(xs, w, b=None, relu_in=True, stem=False):
    """[pool(conv3(relu?(x), w) + b) for x in xs] in one box-sum launch + one strided GEMM launch; check `pool_conv_ok` first.
    stem=True: the pooling is AvgPool3d((1,2,2), stride 2) (time: even frames); else DownSample (2 on every extent > 1)."""
    tmodes = tuple(pool_tmode(x.shape, stem) for x in xs)
    if any(t is None for t in tmodes):
        raise ValueError('pool_conv_group: a member has odd extents (check pool_conv_ok first)')
    return list(PoolConvG.apply(w, b, bool(relu_in), tmodes, *xs))


# ------------------------------------------------------------------------------------------------
# The same kernels in transposed roles: `Upsample(2) -> conv3x3` of the generator's UpBlocks (layers.py:152-195). Nearest
# up-sampling is the box filter of the zero-stuffed map and box filter and convolution commute, so
#     conv3(up2(x)) = boxsum( conv3(zero-stuffed x) )
# and the convolution of a zero-stuffed map only ever multiplies each input pixel by the taps of its parity class: 9 taps per
# INPUT pixel instead of 9 per output pixel — a quarter of the MACs, no up-sampled tensor in memory. Forward = the pooled form's
# data-gradient kernel (class planes) + unbox; data gradient = box-sum + the pooled forward kernel; weight gradient = the pooled
# weight-gradient kernel with its operands swapped (+ a small transpose). First-order only (generator side).
# ------------------------------------------------------------------------------------------------

def up_conv_ok(x, w):
    TF = _TF()
    return (not _DISABLED and not _NO_UPCONV and TF.CONV_PRECISION == 'fp32' and x.dim() == 4 and x.is_cuda and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
            and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0)


class UpConvFn(Function):
    """y = conv3x3(upsample2x(x), w) + b for [N,Cin,h,w] -> [N,Cout,2h,2w]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        N, cin, h, w_ = x.shape
        cout = w.shape[0]
        w5 = w.unsqueeze(2)
        full = (N, cout, 1, 2 * h, 2 * w_)                      # the up-sampled geometry ("full resolution" of the pooled kernels)
        ctx.save_for_backward(x, w)
        ctx.bias = b
        if min(h, w_) < 4:
            # tiny maps: the class planes live on a padded (h+1) x (w+1) grid whose border rows multiply zeros — at 1x1 / 2x2 that is
            # as much work as the plain convolution of the up-sampled map, which has no second pass: forward the plain way (the
            # backward below never needs the up-sampled tensor either way)
            TF = _TF()
            return TF.conv_fwd_raw(TF.Upsample2x.apply(x).unsqueeze(2), w5, b).squeeze(2)
        planes = pool_dgrad_raw([x.unsqueeze(2)], [(N, cin, 1, 2 * h, 2 * w_)], [0], w5, transpose=True)
        y = unbox_raw(planes, [full], [0], scale=1.0, bias=b)[0]
        return y.squeeze(2)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        TF = _TF()
        x, w = ctx.saved_tensors
        b = ctx.bias
        gy = _c(gy)
        N, cin, h, w_ = x.shape
        cout = w.shape[0]
        w5 = w.unsqueeze(2)
        full = (N, cout, 1, 2 * h, 2 * w_)
        gts = boxsum_raw([gy.unsqueeze(2)], [0], False, scale=1.0)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = pool_fwd_raw(gts, [full], [0], w5, None, transpose=True)[0].squeeze(2)
        if ctx.needs_input_grad[1]:
            # operands swapped: "dL/dy" := the small map (Cin channels), "r~" := the box-summed gradient (Cout channels)
            tmp = pool_wgrad_raw(gts, [x.unsqueeze(2)], [full], [0], (cin, cout, 1, 3, 3))

            def swap(out, acc):
                check(lib().t2v_wgrad_swap(_p(tmp), _p(out), cout, cin, 9, 1 if acc else 0, _stream()), 't2v_wgrad_swap')
            done, gw = TF._to_sink(w, swap)
            if not done:
                gw = torch.empty_like(w)
                swap(gw, False)
        if b is not None and ctx.needs_input_grad[2]:
            done, gb = TF._to_sink(b, lambda out, acc: TF.channel_sum_raw(gy.unsqueeze(2), out=out, accum=acc))
            if not done:
                gb = TF.channel_sum_raw(gy.unsqueeze(2))
        return gx, gw, gb


def up_conv(x, w, b=None):
    """conv3x3(upsample2x(x), w) + b without the up-sampled tensor; check `up_conv_ok` first."""
    return UpConvFn.apply(x, w, b)
