"""Grouped small ops of the non-local block and of the multi-level loss / gradient glue: one `t2v_multi` launch per op for
all pyramid levels / members (include/t2v_hip.h, T2V_MJ_*). Part of the autograd surface `txt2vid_amd.functional`, which
re-exports every name defined here; closed under differentiation like the single-tensor Functions."""
import torch
from torch.autograd import Function

from ._ops import lib, check, _c, _p, _stream


# ------------------------------------------------------------------------------------------------
# the non-local block's small ops over several tensors at once (`t2v_multi`): one launch per op for all pyramid
# levels / members instead of one per level. Same closure under differentiation as the single-tensor Functions;
# members that receive no gradient (None) are left out of the backward launches.
# ------------------------------------------------------------------------------------------------
(MJ_SCALE, MJ_SCALE_ADD, MJ_DOT, MJ_MAXPOOL, MJ_MAXSCATTER, MJ_MAXGATHER, MJ_SOFTMAX, MJ_SOFTMAX_BWD, MJ_SOFTMAX_BWD_BWD_Y, MJ_BMM,
 MJ_RELU_MASK, MJ_ROWSUM, MJ_ROWBCAST, MJ_ADD, MJ_CATLERP, MJ_CATCOLS, MJ_SLICECOLS, MJ_EMBEDCOLS) = range(1, 19)


def _mj(op, jobs, scalar=None, dot_out=None, dot_accum=False):
    """jobs: list of dicts with keys a b c out out2 (tensors) and n d0 d1 d2 f0 f1 (ints)."""
    from ._lib import MultiJob
    for at in range(0, len(jobs), 8):
        part = jobs[at:at + 8]
        arr = (MultiJob * len(part))()
        for a, q in zip(arr, part):
            for key in ('a', 'b', 'c', 'out', 'out2'):
                t = q.get(key)
                setattr(a, key, t.data_ptr() if t is not None else None)
            a.n = q['n']
            for key in ('d0', 'd1', 'd2', 'f0', 'f1'):
                setattr(a, key, q.get(key, 0))
        ws = None
        if op == MJ_DOT:
            arr[0].out, arr[0].f0 = dot_out.data_ptr(), (1 if (at > 0 or dot_accum) else 0)
            ws = torch.empty((int(lib().t2v_multi_ws_floats(op, arr, len(part))),), device=dot_out.device, dtype=torch.float32)
        check(lib().t2v_multi(op, arr, len(part), _p(scalar), _p(ws), _stream()), 't2v_multi')


def _live(gs):
    return [i for i, g in enumerate(gs) if g is not None]


class AddG(Function):
    """ys[i] = as[i] + bs[i] for n pairs of same-shaped tensors in ONE launch (args: a_0..a_{n-1}, b_0..b_{n-1})."""

    @staticmethod
    def forward(ctx, *ts):
        n = len(ts) // 2
        a, b = [_c(t) for t in ts[:n]], [_c(t) for t in ts[n:]]
        ys = [torch.empty_like(t) for t in a]
        _mj(MJ_ADD, [dict(a=x, b=y, out=o, n=x.numel()) for x, y, o in zip(a, b, ys)])
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        if torch.is_grad_enabled() and any(g is not None and g.requires_grad for g in gs):
            # recorded backward (the gradient penalty's sweep): dL/dy feeds BOTH summands' adjoints — a grouped fork, so that the
            # second backward sums the two contributions of all members in one launch (not one engine add per member)
            a, b = _fork_some(list(gs))
            return tuple(a) + tuple(b)
        return tuple(gs) + tuple(gs)


class ForkG(Function):
    """Two autograd aliases of each of n tensors: `xs -> (xs', xs'')` for an activation list that feeds two consumers. The
    values are views (no kernel); the adjoint adds the two gradients of ALL members in one launch (`AddG`) — without the
    fork the autograd engine accumulates them with one ATen add per member. Closed under differentiation (AddG's adjoint
    hands its gradient to both summands)."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs) + tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        n = len(gs) // 2
        out = [None] * n
        both = [i for i in range(n) if gs[i] is not None and gs[n + i] is not None]
        for i in range(n):
            if i not in both:
                out[i] = gs[i] if gs[i] is not None else gs[n + i]
        if both:
            sums = AddG.apply(*([gs[i] for i in both] + [gs[n + i] for i in both]))
            for i, t in zip(both, sums):
                out[i] = t
        return tuple(out)


def fork_group(xs):
    """(xs', xs''): two aliases of every tensor of `xs` whose gradients are summed in one launch. Tensors that need no
    gradient are passed through as they are (a fork of them would only mark their aliases as requiring one)."""
    xs = list(xs)
    live = [i for i, x in enumerate(xs) if x.requires_grad]
    if not live or not torch.is_grad_enabled():
        return xs, xs
    res = ForkG.apply(*[xs[i] for i in live])
    a, b = list(xs), list(xs)
    for k, i in enumerate(live):
        a[i], b[i] = res[k], res[len(live) + k]
    return a, b


def _fork_some(ts):
    """`fork_group` over a list that may hold None entries (members without a gradient)."""
    idx = [i for i, t in enumerate(ts) if t is not None]
    a, b = list(ts), list(ts)
    if idx:
        fa, fb = fork_group([ts[i] for i in idx])
        for k, i in enumerate(idx):
            a[i], b[i] = fa[k], fb[k]
    return a, b


class CatColsG(Function):
    """outs[i] = torch.cat((as[i], bs[i]), 1) for n pairs of 2-D tensors in ONE launch (the conditional heads' feature || caption
    concatenation of every pyramid level, resnet3d.py:53). args: n, a_0.., b_0..  Closed under differentiation (SliceColsG /
    EmbedColsG)."""

    @staticmethod
    def forward(ctx, n, *ts):
        a, b = [_c(t) for t in ts[:n]], [_c(t) for t in ts[n:]]
        outs = [torch.empty((x.shape[0], x.shape[1] + y.shape[1]), device=x.device, dtype=torch.float32) for x, y in zip(a, b)]
        _mj(MJ_CATCOLS, [dict(a=x, b=y, out=o, n=x.shape[0], d0=x.shape[1], d1=y.shape[1]) for x, y, o in zip(a, b, outs)])
        ctx.cfg = (n, [(x.shape[1], y.shape[1]) for x, y in zip(a, b)])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, widths = ctx.cfg
        ga, gb = [None] * n, [None] * n
        la = [i for i in _live(gs) if ctx.needs_input_grad[1 + i]]
        lb = [i for i in _live(gs) if ctx.needs_input_grad[1 + n + i]]
        if la:
            for i, r in zip(la, SliceColsG.apply(tuple((sum(widths[i]), 0, widths[i][0]) for i in la), *[gs[i] for i in la])):
                ga[i] = r
        if lb:
            for i, r in zip(lb, SliceColsG.apply(tuple((sum(widths[i]), widths[i][0], widths[i][1]) for i in lb), *[gs[i] for i in lb])):
                gb[i] = r
        return (None,) + tuple(ga) + tuple(gb)


class SliceColsG(Function):
    """outs[i] = xs[i][:, off:off+n] with cfgs[i] = (total, off, n)."""

    @staticmethod
    def forward(ctx, cfgs, *xs):
        xs = [_c(x) for x in xs]
        outs = [torch.empty((x.shape[0], c[2]), device=x.device, dtype=torch.float32) for x, c in zip(xs, cfgs)]
        _mj(MJ_SLICECOLS, [dict(a=x, out=o, n=x.shape[0], d0=c[0], d1=c[1], d2=c[2]) for x, o, c in zip(xs, outs, cfgs)])
        ctx.cfg = cfgs
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        cfgs = ctx.cfg
        live = _live(gs)
        out = [None] * len(cfgs)
        if live:
            for i, r in zip(live, EmbedColsG.apply(tuple(cfgs[i] for i in live), *[gs[i] for i in live])):
                out[i] = r
        return (None,) + tuple(out)


class EmbedColsG(Function):
    """outs[i] = zeros [rows, total] with gs[i] written at columns off:off+n (the adjoint of SliceColsG)."""

    @staticmethod
    def forward(ctx, cfgs, *gs):
        gs = [_c(g) for g in gs]
        outs = [torch.empty((g.shape[0], c[0]), device=g.device, dtype=torch.float32) for g, c in zip(gs, cfgs)]
        _mj(MJ_EMBEDCOLS, [dict(a=g, out=o, n=g.shape[0], d0=c[0], d1=c[1], d2=c[2]) for g, o, c in zip(gs, outs, cfgs)])
        ctx.cfg = cfgs
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *ggs):
        cfgs = ctx.cfg
        live = _live(ggs)
        out = [None] * len(cfgs)
        if live:
            for i, r in zip(live, SliceColsG.apply(tuple(cfgs[i] for i in live), *[ggs[i] for i in live])):
                out[i] = r
        return (None,) + tuple(out)


class HeadRowsG(Function):
    """outs[i] = xs[i][:ns[i]] (views) for n dense tensors; the adjoint writes the zero-padded gradients of ALL members in one
    launch (`EmbedColsG` on the flattened tensors) — the native slices' adjoint is a zero-fill + a copy per member."""

    @staticmethod
    def forward(ctx, ns, *xs):
        xs = [_c(x) for x in xs]
        ctx.cfg = (tuple(int(n) for n in ns), tuple(tuple(x.shape) for x in xs))
        ctx.set_materialize_grads(False)
        return tuple(x[:n] for x, n in zip(xs, ns))

    @staticmethod
    def backward(ctx, *gs):
        ns, shapes = ctx.cfg
        out = [None] * len(ns)
        live = [i for i in _live(gs) if ctx.needs_input_grad[1 + i]]
        if live:
            cfgs = []
            for i in live:
                total = 1
                for d in shapes[i]:
                    total *= d
                cfgs.append((total, 0, total // shapes[i][0] * ns[i]))
            res = EmbedColsG.apply(tuple(cfgs), *[gs[i].reshape(1, -1) for i in live])
            for i, r in zip(live, res):
                out[i] = r.view(shapes[i])
        return (None,) + tuple(out)


class SplitRowsG(Function):
    """(xs[i][:ns[i]], xs[i][ns[i]:]) as views for n dense tensors — outputs: the n heads, then the n tails. The adjoint writes
    both gradients of ALL members into their buffers in one launch (`CatColsG` on the flattened halves) where the per-member
    form costs two copies each (the [real; fake] logits of every pyramid level in the D step)."""

    @staticmethod
    def forward(ctx, ns, *xs):
        xs = [_c(x) for x in xs]
        ctx.cfg = (tuple(int(n) for n in ns), tuple(tuple(x.shape) for x in xs))
        ctx.set_materialize_grads(False)
        return tuple(x[:n] for x, n in zip(xs, ns)) + tuple(x[n:] for x, n in zip(xs, ns))

    @staticmethod
    def backward(ctx, *gs):
        ns, shapes = ctx.cfg
        n = len(ns)
        out = [None] * n
        both = [i for i in range(n) if gs[i] is not None and gs[n + i] is not None]
        if both:
            res = CatColsG.apply(len(both), *([gs[i].reshape(1, -1) for i in both] + [gs[n + i].reshape(1, -1) for i in both]))
            for i, r in zip(both, res):
                out[i] = r.view(shapes[i])
        for i in range(n):
            if i in both or (gs[i] is None and gs[n + i] is None):
                continue
            rows = shapes[i][0]
            total = 1
            for d in shapes[i]:
                total *= d
            per = total // rows
            g, off, cnt = (gs[i], 0, ns[i] * per) if gs[i] is not None else (gs[n + i], ns[i] * per, (rows - ns[i]) * per)
            out[i] = EmbedColsG.apply(((total, off, cnt),), g.reshape(1, -1))[0].view(shapes[i])
        return (None,) + tuple(out)


def split_rows_group(xs, ns):
    """([x[:n] ...], [x[n:] ...]) with ONE launch in the adjoint for all members (at most 8 per launch)."""
    xs = list(xs)
    if not torch.is_grad_enabled() or not any(x.requires_grad for x in xs) or len(xs) > 8:
        return [x[:n] for x, n in zip(xs, ns)], [x[n:] for x, n in zip(xs, ns)]
    res = SplitRowsG.apply(tuple(ns), *xs)
    return list(res[:len(xs)]), list(res[len(xs):])


def head_rows_group(xs, ns):
    """[x[:n] for x, n in zip(xs, ns)] with ONE launch in the adjoint for all members (at most 8 per launch)."""
    xs = list(xs)
    if not torch.is_grad_enabled() or not any(x.requires_grad for x in xs) or len(xs) > 8:
        return [x[:n] for x, n in zip(xs, ns)]
    return list(HeadRowsG.apply(tuple(ns), *xs))


def cat_features_group(as_, bs):
    """[torch.cat((a, b), 1) for a, b in zip(as_, bs)] in one launch (2-D tensors, at most 8 pairs per launch)."""
    as_, bs = list(as_), list(bs)
    return list(CatColsG.apply(len(as_), *(as_ + bs)))


class MaxPool2x2G(Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(x) for x in xs]
        ys = [torch.empty(tuple(x.shape[:-2]) + (x.shape[-2] // 2, x.shape[-1] // 2), device=x.device, dtype=torch.float32) for x in xs]
        idxs = [torch.empty(y.shape, device=y.device, dtype=torch.int32) for y in ys]
        _mj(MJ_MAXPOOL, [dict(a=x, out=y, out2=i, n=x.numel() // (x.shape[-2] * x.shape[-1]), d0=x.shape[-2], d1=x.shape[-1])
                         for x, y, i in zip(xs, ys, idxs)])
        ctx.save_for_backward(*idxs)
        ctx.in_shapes = [tuple(x.shape) for x in xs]
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        idxs = ctx.saved_tensors
        live = _live(gs)
        out = [None] * len(gs)
        if live:
            res = MaxScatterG.apply(tuple(ctx.in_shapes[i] for i in live), len(live), *([gs[i] for i in live] + [idxs[i] for i in live]))
            for i, r in zip(live, res):
                out[i] = r
        return tuple(out)


class MaxScatterG(Function):
    @staticmethod
    def forward(ctx, in_shapes, n, *gs_idxs):
        gs, idxs = [_c(g) for g in gs_idxs[:n]], gs_idxs[n:]
        gxs = [torch.empty(sh, device=g.device, dtype=torch.float32) for sh, g in zip(in_shapes, gs)]
        _mj(MJ_MAXSCATTER, [dict(a=g, b=i, out=gx, n=gx.numel() // (sh[-2] * sh[-1]), d0=sh[-2], d1=sh[-1])
                            for g, i, gx, sh in zip(gs, idxs, gxs, in_shapes)])
        ctx.save_for_backward(*idxs)
        ctx.cfg = (in_shapes, n)
        ctx.set_materialize_grads(False)
        return tuple(gxs)

    @staticmethod
    def backward(ctx, *ggs):
        idxs = ctx.saved_tensors
        in_shapes, n = ctx.cfg
        live = _live(ggs)
        out = [None] * n
        if live:
            res = MaxGatherG.apply(tuple(in_shapes[i] for i in live), len(live), *([ggs[i] for i in live] + [idxs[i] for i in live]))
            for i, r in zip(live, res):
                out[i] = r
        return (None, None) + tuple(out) + (None,) * n


class MaxGatherG(Function):
    @staticmethod
    def forward(ctx, in_shapes, n, *xs_idxs):
        xs, idxs = [_c(x) for x in xs_idxs[:n]], xs_idxs[n:]
        ys = [torch.empty(tuple(sh[:-2]) + (sh[-2] // 2, sh[-1] // 2), device=x.device, dtype=torch.float32) for sh, x in zip(in_shapes, xs)]
        _mj(MJ_MAXGATHER, [dict(a=x, b=i, out=y, n=x.numel() // (sh[-2] * sh[-1]), d0=sh[-2], d1=sh[-1])
                           for x, i, y, sh in zip(xs, idxs, ys, in_shapes)])
        ctx.save_for_backward(*idxs)
        ctx.cfg = (in_shapes, n)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        idxs = ctx.saved_tensors
        in_shapes, n = ctx.cfg
        live = _live(gs)
        out = [None] * n
        if live:
            res = MaxScatterG.apply(tuple(in_shapes[i] for i in live), len(live), *([gs[i] for i in live] + [idxs[i] for i in live]))
            for i, r in zip(live, res):
                out[i] = r
        return (None, None) + tuple(out) + (None,) * n


def _bmm_dims(A, B, ta, tb):
    M = A.shape[2] if ta else A.shape[1]
    K = A.shape[1] if ta else A.shape[2]
    N = B.shape[1] if tb else B.shape[2]
    return M, N, K


class BmmG(Function):
    """Cs[i] = op(As[i]) @ op(Bs[i]) (batched, per-member transposes cfgs[i] = (ta, tb)) in one launch."""

    @staticmethod
    def forward(ctx, cfgs, n, *ABs):
        # 4n tensors: the last 2n are autograd ALIASES of the operands (ForkG), saved for the adjoint in their place — a recorded
        # adjoint (the gradient penalty's sweep) then hangs its products on the aliases, and the fork sums the operands' two
        # gradient contributions for all members in one launch (otherwise: one engine add per operand and member)
        aliased = len(ABs) == 4 * n
        As, Bs = [_c(t) for t in ABs[:n]], [_c(t) for t in ABs[n:2 * n]]
        jobs, Cs = [], []
        for A, B, (ta, tb) in zip(As, Bs, cfgs):
            M, N, K = _bmm_dims(A, B, ta, tb)
            Cm = torch.empty((A.shape[0], M, N), device=A.device, dtype=torch.float32)
            Cs.append(Cm)
            jobs.append(dict(a=A, b=B, out=Cm, n=A.shape[0], d0=M, d1=N, d2=K, f0=int(ta), f1=int(tb)))
        _mj(MJ_BMM, jobs)
        if aliased:
            ctx.save_for_backward(*[_c(t) for t in ABs[2 * n:]])
        else:
            ctx.save_for_backward(*As, *Bs)
        ctx.cfg = (cfgs, n)
        ctx.aliased = aliased
        ctx.set_materialize_grads(False)
        return tuple(Cs)

    @staticmethod
    def backward(ctx, *Gs):
        cfgs, n = ctx.cfg
        saved = ctx.saved_tensors
        As, Bs = saved[:n], saved[n:]
        dAs, dBs = [None] * n, [None] * n
        la = [i for i in _live(Gs) if ctx.needs_input_grad[2 + i]]
        lb = [i for i in _live(Gs) if ctx.needs_input_grad[2 + n + i]]
        Gb = Gs
        if la and lb and torch.is_grad_enabled():     # recorded backward: dL/dC feeds both products — grouped fork
            Gs, Gb = _fork_some(list(Gs))
        if la:        # dA = G @ B'^T (ta = 0)  |  B' @ G^T (ta = 1)
            X = [Gs[i] if not cfgs[i][0] else Bs[i] for i in la]
            Y = [Bs[i] if not cfgs[i][0] else Gs[i] for i in la]
            c = tuple((False, not cfgs[i][1]) if not cfgs[i][0] else (cfgs[i][1], True) for i in la)
            for i, r in zip(la, BmmG.apply(c, len(la), *(X + Y))):
                dAs[i] = r
        if lb:        # dB = A'^T @ G (tb = 0)  |  G^T @ A' (tb = 1)
            X = [As[i] if not cfgs[i][1] else Gb[i] for i in lb]
            Y = [Gb[i] if not cfgs[i][1] else As[i] for i in lb]
            c = tuple((not cfgs[i][0], False) if not cfgs[i][1] else (True, cfgs[i][0]) for i in lb)
            for i, r in zip(lb, BmmG.apply(c, len(lb), *(X + Y))):
                dBs[i] = r
        return (None, None) + tuple(dAs) + tuple(dBs) + ((None,) * (2 * n) if ctx.aliased else ())


class SoftmaxG(Function):
    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(x) for x in xs]
        ys = [torch.empty_like(x) for x in xs]
        _mj(MJ_SOFTMAX, [dict(a=x, out=y, n=x.numel() // x.shape[-1], d0=x.shape[-1]) for x, y in zip(xs, ys)])
        # outputs: ys for the consumers, then n ALIASES of them that the adjoint saves: a recorded adjoint (gradient penalty) hangs
        # its use of y on the alias, and this node sums the two gradients of all members in one launch (AddG) itself
        al = [y.view_as(y) for y in ys]
        ctx.save_for_backward(*al)
        ctx.set_materialize_grads(False)
        return tuple(ys) + tuple(al)

    @staticmethod
    def backward(ctx, *gs2):
        ys = ctx.saved_tensors
        n = len(ys)
        gs, g2 = list(gs2[:n]), gs2[n:]
        both = [i for i in range(n) if gs[i] is not None and g2[i] is not None]
        for i in range(n):
            if gs[i] is None:
                gs[i] = g2[i]
        if both:
            for i, t in zip(both, AddG.apply(*([gs[i] for i in both] + [g2[i] for i in both]))):
                gs[i] = t
        live = _live(gs)
        out = [None] * len(gs)
        if live:
            for i, r in zip(live, SoftmaxBwdG.apply(len(live), *([ys[i] for i in live] + [gs[i] for i in live]))):
                out[i] = r
        return tuple(out)


class SoftmaxBwdG(Function):
    @staticmethod
    def forward(ctx, n, *ys_gys):
        ys, gys = [_c(t) for t in ys_gys[:n]], [_c(t) for t in ys_gys[n:]]
        gxs = [torch.empty_like(y) for y in ys]
        _mj(MJ_SOFTMAX_BWD, [dict(a=y, b=g, out=o, n=y.numel() // y.shape[-1], d0=y.shape[-1]) for y, g, o in zip(ys, gys, gxs)])
        ctx.save_for_backward(*ys, *gys)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(gxs)

    @staticmethod
    def backward(ctx, *ggs):
        n = ctx.n
        saved = ctx.saved_tensors
        ys, gys = saved[:n], saved[n:]
        live = _live(ggs)
        d_ys, d_gys = [None] * n, [None] * n
        if live:
            qq = [_c(ggs[i]) for i in live]
            ly = [i for i in live if ctx.needs_input_grad[1 + i]]
            if ly:
                outs = [torch.empty_like(ys[i]) for i in ly]
                _mj(MJ_SOFTMAX_BWD_BWD_Y, [dict(a=ys[i], b=gys[i], c=_c(ggs[i]), out=o, n=ys[i].numel() // ys[i].shape[-1],
                                                d0=ys[i].shape[-1]) for i, o in zip(ly, outs)])
                for i, o in zip(ly, outs):
                    d_ys[i] = o
            lg = [i for i in live if ctx.needs_input_grad[1 + n + i]]
            if lg:
                for i, r in zip(lg, SoftmaxBwdG.apply(len(lg), *([ys[i] for i in lg] + [ggs[i] for i in lg]))):
                    d_gys[i] = r
            del qq
        return (None,) + tuple(d_ys) + tuple(d_gys)


class ScaleG(Function):
    """ys[i] = s * as[i] with s a 0-d device tensor."""

    @staticmethod
    def forward(ctx, s, *as_):
        as_ = [_c(a) for a in as_]
        ys = [torch.empty_like(a) for a in as_]
        _mj(MJ_SCALE, [dict(a=a, out=y, n=a.numel()) for a, y in zip(as_, ys)], scalar=s)
        ctx.save_for_backward(s, *as_)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        s, as_ = saved[0], saved[1:]
        live = _live(gs)
        d_s, d_as = None, [None] * len(as_)
        if live:
            lg = lg2 = [gs[i] for i in live]
            if ctx.needs_input_grad[0] and torch.is_grad_enabled():      # recorded backward: the gradients feed the dot and the scale
                lg, lg2 = fork_group(lg)
            if ctx.needs_input_grad[0]:
                done, d_s = _dot_to_sink(s, lg2, [as_[i] for i in live])
                if not done:
                    d_s = DotG.apply(len(live), *(lg2 + [as_[i] for i in live]))
            for i, r in zip(live, ScaleG.apply(s, *lg)):
                d_as[i] = r
        return (d_s,) + tuple(d_as)


def _dot_to_sink(s, as_, bs):
    """sum_i <as[i], bs[i]> written straight into the gradient-sink slot of the 0-d parameter `s` (the non-local block's gain),
    accumulating when the slot already holds this step's first contribution: (handled, gradient for autograd | None). The
    engine otherwise adds the forward graph's and the gradient penalty's contributions with an ATen launch of its own."""
    from . import functional as TF
    if torch.is_grad_enabled():
        return False, None
    as_, bs = [_c(t) for t in as_], [_c(t) for t in bs]

    def compute(out, accum):
        _mj(MJ_DOT, [dict(a=a, b=b, out=out, n=a.numel()) for a, b in zip(as_, bs)], dot_out=out, dot_accum=accum)
    return TF._to_sink(s, compute)


class DotG(Function):
    """sum_i <as[i], bs[i]> as one 0-d tensor."""

    @staticmethod
    def forward(ctx, n, *as_bs):
        as_, bs = [_c(t) for t in as_bs[:n]], [_c(t) for t in as_bs[n:]]
        out = torch.empty((), device=as_[0].device, dtype=torch.float32)
        _mj(MJ_DOT, [dict(a=a, b=b, out=out, n=a.numel()) for a, b in zip(as_, bs)], dot_out=out)
        ctx.save_for_backward(*as_, *bs)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        as_, bs = saved[:n], saved[n:]
        if g is None:
            return (None,) * (1 + 2 * n)
        return (None,) + tuple(ScaleG.apply(g, *bs)) + tuple(ScaleG.apply(g, *as_))


class ScaleAddG(Function):
    """ys[i] = s * os[i] + xs[i]  (layers.py:36,68 over all levels)."""

    @staticmethod
    def forward(ctx, s, n, *os_xs):
        os_, xs = [_c(t) for t in os_xs[:n]], [_c(t) for t in os_xs[n:]]
        ys = [torch.empty_like(o) for o in os_]
        _mj(MJ_SCALE_ADD, [dict(a=o, b=x, out=y, n=o.numel()) for o, x, y in zip(os_, xs, ys)], scalar=s)
        ctx.save_for_backward(s, *os_)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        saved = ctx.saved_tensors
        s, os_ = saved[0], saved[1:]
        live = _live(gs)
        d_s, d_os = None, [None] * n
        res = gs
        if live:
            gd = gsc = gs
            if torch.is_grad_enabled() and any(gs[i].requires_grad for i in live):
                # recorded backward: dL/dy feeds the dot (d gamma), the scale (d o) and the residual — grouped forks, see AddG
                gsc, res = _fork_some(list(gs))
                gd = gsc
                if ctx.needs_input_grad[0]:
                    gd, gsc = _fork_some(list(gsc))
            if ctx.needs_input_grad[0]:
                done, d_s = _dot_to_sink(s, [gd[i] for i in live], [os_[i] for i in live])
                if not done:
                    d_s = DotG.apply(len(live), *([gd[i] for i in live] + [os_[i] for i in live]))
            lo = [i for i in live if ctx.needs_input_grad[2 + i]]
            if lo:
                for i, r in zip(lo, ScaleG.apply(s, *[gsc[i] for i in lo])):
                    d_os[i] = r
        return (d_s, None) + tuple(d_os) + tuple(res)


class ReluMaskG(Function):
    """outs[i] = gs[i] * [xs[i] > 0] for several tensors in one launch (linear in g; its adjoint is itself)."""

    @staticmethod
    def forward(ctx, n, *gs_xs):
        gs, xs = [_c(t) for t in gs_xs[:n]], [_c(t) for t in gs_xs[n:]]
        outs = [torch.empty_like(g) for g in gs]
        _mj(MJ_RELU_MASK, [dict(a=g, b=x, out=o, n=g.numel()) for g, x, o in zip(gs, xs, outs)])
        ctx.save_for_backward(*xs)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *ggs):
        xs = ctx.saved_tensors
        live = _live(ggs)
        out = [None] * len(xs)
        if live:
            for i, r in zip(live, ReluMaskG.apply(len(live), *([ggs[i] for i in live] + [xs[i] for i in live]))):
                out[i] = r
        return (None,) + tuple(out) + (None,) * len(xs)


class RowSumG(Function):
    """ys[i] = xs[i] summed over everything behind the first two dims ([b,C,...] -> [b,C]) for several tensors."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(x) for x in xs]
        ys = [torch.empty(tuple(x.shape[:2]), device=x.device, dtype=torch.float32) for x in xs]
        _mj(MJ_ROWSUM, [dict(a=x, out=y, n=y.numel(), d0=x.numel() // y.numel()) for x, y in zip(xs, ys)])
        ctx.shapes = [tuple(x.shape) for x in xs]
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        live = _live(gs)
        out = [None] * len(gs)
        if live:
            for i, r in zip(live, RowBcastG.apply(tuple(ctx.shapes[i] for i in live), *[gs[i] for i in live])):
                out[i] = r
        return tuple(out)


class RowBcastG(Function):
    @staticmethod
    def forward(ctx, shapes, *gs):
        gs = [_c(g) for g in gs]
        outs = [torch.empty(sh, device=g.device, dtype=torch.float32) for sh, g in zip(shapes, gs)]
        _mj(MJ_ROWBCAST, [dict(a=g, out=o, n=g.numel(), d0=o.numel() // g.numel()) for g, o in zip(gs, outs)])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *ggs):
        live = _live(ggs)
        out = [None] * len(ggs)
        if live:
            for i, r in zip(live, RowSumG.apply(*[ggs[i] for i in live])):
                out[i] = r
        return (None,) + tuple(out)


def dot_group(as_, bs):
    """sum_i <as[i], bs[i]> over n pairs of same-shaped tensors as one 0-d tensor (`DotG`: one launch + its final reduce)."""
    as_, bs = list(as_), list(bs)
    return DotG.apply(len(as_), *(as_ + bs))


def sum_spatial_group(xs):
    return list(RowSumG.apply(*xs))


def max_pool2x2_group(xs):
    return list(MaxPool2x2G.apply(*xs))


def bmm_group(As, Bs, ta, tb):
    As, Bs = list(As), list(Bs)
    cfgs = tuple((ta, tb) for _ in As)
    if torch.is_grad_enabled() and all(t.requires_grad for t in As + Bs):
        # operands + their aliases for the adjoint (see BmmG.forward): one grouped fork for all of them
        ops, als = fork_group(As + Bs)
        return list(BmmG.apply(cfgs, len(As), *(ops + als)))
    return list(BmmG.apply(cfgs, len(As), *(As + Bs)))


def softmax_lastdim_group(xs):
    return list(SoftmaxG.apply(*xs))[:len(xs)]


def scale_add_group(gamma, os_, xs):
    return list(ScaleAddG.apply(gamma, len(os_), *(list(os_) + list(xs))))
