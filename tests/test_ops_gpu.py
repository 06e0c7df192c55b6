"""GPU: every C-ABI kernel (through txt2vid_amd.functional) against plain fp32 PyTorch CPU ops.
Tolerances: fp32 MFMA is an exact fp32 fma chain; differences come from summation order only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


def rnd(seed, *shape):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


def close(a, b, rtol=1e-4, atol=1e-4):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale)


def ref_conv(x, w, b):
    if x.dim() == 5:
        return F.conv3d(x, w, b, padding=tuple(k // 2 for k in w.shape[2:]))
    if x.dim() == 4:
        return F.conv2d(x, w, b, padding=tuple(k // 2 for k in w.shape[2:]))
    return F.linear(x, w, b)


from conv_cases import SINGLE_CASES as CONV_CASES, GROUPED_CASES, BF16_CASES      # shared with the CPU launch-plan coverage test


@pytest.mark.parametrize('xs,cout,k', CONV_CASES)
def test_conv_fwd_bwd(xs, cout, k):
    from txt2vid_amd import functional as TF
    x = rnd(1, *xs)
    w = rnd(2, cout, xs[1], *k) * 0.1
    b = rnd(3, cout)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ref_conv(xr, wr, br)
    gy = rnd(4, *yr.shape)
    (yr * gy).sum().backward()
    xd = x.to(dev()).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev()))
    bd = torch.nn.Parameter(b.to(dev()))
    yd = TF.conv(xd, wd, bd)
    close(yd, yr)
    (yd * gy.to(dev())).sum().backward()
    close(xd.grad, xr.grad)
    close(wd.grad, wr.grad, rtol=2e-4, atol=2e-4)
    close(bd.grad, br.grad, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('case', GROUPED_CASES, ids=[c[0] for c in GROUPED_CASES])
def test_conv_grouped_at_benchmark_size(case):
    """The discriminator's grouped launches at the size bench.py runs them (BASELINE configs[1], per-GPU batch 32: the 8
    members of a D-step pass, M = 393 216 voxels for the 64->64 stem convolution): forward (+ fused input ReLU, bias), the
    data gradient with the ReLU adjoint in its epilogue, and the weight + bias gradient summed over all members — the
    256x64 strip GEMM and the 171-way split weight gradient with its many-splits reduce — against torch on the CPU.
    Per-member references in fp32; the weight gradient (a sum over up to 393 216 voxels) against an fp64 accumulation of
    the per-member fp32 references."""
    from txt2vid_amd import functional as TF
    name, cin, cout, k, members, relu_in = case
    w = rnd(2, cout, cin, *k) * (1.0 / np.sqrt(cin * 27.0))
    b = rnd(3, cout) * 0.1
    xs = [rnd(10 + i, n, cin, d, h, wd) for i, (n, d, h, wd) in enumerate(members)]
    gys = [rnd(40 + i, n, cout, d, h, wd) for i, (n, d, h, wd) in enumerate(members)]
    wd_, bd_ = w.to(dev()), b.to(dev())
    xd, gyd = [x.to(dev()) for x in xs], [g.to(dev()) for g in gys]
    ys = TF.conv_group_raw(xd, wd_, bd_, relu_in, 0)
    gxs = TF.conv_group_raw(gyd, wd_, None, False, 1, masks=xd if relu_in else None)
    dbias = torch.empty(cout, device=dev())
    dw = TF.conv_group_wgrad_raw(xd, gyd, tuple(w.shape), relu_in, dbias=dbias)
    torch.cuda.synchronize()
    dw_ref = torch.zeros(w.shape, dtype=torch.float64)
    db_ref = torch.zeros(cout, dtype=torch.float64)
    for i, (x, gy) in enumerate(zip(xs, gys)):
        xr = x.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv3d(F.relu(xr) if relu_in else xr, wr, br, padding=1)
        close(ys[i], yr)
        (yr * gy).sum().backward()
        close(gxs[i], xr.grad)
        dw_ref += wr.grad.double()
        db_ref += br.grad.double()
    close(dw, dw_ref, rtol=2e-4, atol=2e-4)
    close(dbias, db_ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('cin,cout,k,members', [
    (64, 64, (3, 3, 3), [(4, 4, 16, 16), (2, 2, 32, 32)]),          # 3-tap rows, many splits of a small weight (kind 1)
    (128, 256, (3, 3, 3), [(4, 2, 4, 4), (2, 1, 8, 8)]),             # 3-tap rows, one workgroup per 64 pairs (kind 0)
    (64, 96, (1, 1, 1), [(3, 4, 8, 8)]),                             # per-tap kernel
    (16, 32, (3, 3, 3), [(2, 4, 6, 6)]),                             # Cin < 64: (tap, ci) column tiles, bias side sums in the first column tile
])
def test_deferred_weight_gradient_reduce(cin, cout, k, members):
    """`GradSink` leaves each weight-gradient launch's k-split slab pending and sums all of them in ONE
    `t2v_wgrad_reduce_multi` launch: same numbers as the per-launch reduce — bit for bit for a single source; three
    sources of one parameter (first-order + gradient-penalty terms) summed in order; an immediate producer in between
    flushes first (ordering)."""
    from txt2vid_amd import functional as TF
    from txt2vid_amd.dist import GradArena
    w = torch.nn.Parameter(rnd(2, cout, cin, *k).to(dev()) * 0.1)
    b = torch.nn.Parameter(rnd(3, cout).to(dev()))
    sets = []
    for r in range(3):
        xs = [rnd(10 + 7 * r + i, n, cin, d, h, wd).to(dev()) for i, (n, d, h, wd) in enumerate(members)]
        gys = [rnd(40 + 7 * r + i, n, cout, d, h, wd).to(dev()) for i, (n, d, h, wd) in enumerate(members)]
        sets.append((xs, gys))
    # reference: immediate reduces, accumulated launch by launch
    dw_ref, db_ref = torch.empty_like(w), torch.empty_like(b)
    for r, (xs, gys) in enumerate(sets):
        TF.conv_group_wgrad_raw(xs, gys, tuple(w.shape), True, out=dw_ref, accum=r > 0, dbias=db_ref, accum_bias=r > 0)
    dw1 = TF.conv_group_wgrad_raw(sets[0][0], sets[0][1], tuple(w.shape), True)
    arena = GradArena([w, b])
    sink = TF.GradSink([arena])
    assert sink.defer
    old = TF.set_grad_sink(sink)
    try:
        with torch.no_grad():
            # one source: bit-identical to the immediate path
            TF.grad_sink_reset()
            done, gw = TF._to_sink_w(w, sets[0][0], sets[0][1], True)
            assert done and gw is not None and sink.pending
            TF.grad_sink_flush()
            assert not sink.pending
            assert torch.equal(gw, dw1)
            # three sources of the same weight (+ bias through the fused side sums when Cin >= 64)
            TF.grad_sink_reset()
            for r, (xs, gys) in enumerate(sets):
                done, gw_r, gb_r = TF._to_sink_wb(w, b, xs, gys, True)
                assert done and (gw_r is None) == (r > 0)
            TF.grad_sink_flush()
            views = arena.views()
            close(views[0], dw_ref, rtol=1e-5, atol=1e-5)
            close(views[1], db_ref, rtol=1e-5, atol=1e-5)
            # an immediate producer of the same parameter while a slab is pending: the pending part lands first
            TF.grad_sink_reset()
            TF._to_sink_w(w, sets[0][0], sets[0][1], True)
            extra = rnd(99, *w.shape).to(dev())
            done, _ = TF._to_sink(w, lambda out, acc: TF.copy_into(out + extra, out) if acc else TF.copy_into(extra, out))
            assert done and not sink.pending
            TF._to_sink_w(w, sets[1][0], sets[1][1], True)
            TF.grad_sink_flush()
            want = dw1 + extra + TF.conv_group_wgrad_raw(sets[1][0], sets[1][1], tuple(w.shape), True)
            close(arena.views()[0], want, rtol=1e-5, atol=1e-5)
    finally:
        TF.set_grad_sink(old)


@pytest.mark.parametrize('b,N,Nk', [(3, 1024, 256), (2, 900, 225), (1, 64, 16), (2, 4096, 1024)])
def test_fused_nonlocal_attend(b, N, Nk):
    """`t2v_nonlocal_fwd/_bwd` (scores -> softmax -> weighted sum with beta kept in registers) against torch on the CPU:
    o = g . softmax(theta^T phi)^T and the three input gradients, at the generator's head sizes (4 / 16), for map sizes
    with partial key / query tiles and the 64x64 map of BASELINE configs[4] (beta would be 4096 x 1024 per frame)."""
    from txt2vid_amd import functional as TF
    th, ph, g = rnd(1, b, 4, N), rnd(2, b, 4, Nk), rnd(3, b, 16, Nk)
    go = rnd(4, b, 16, N)
    ref = [t.clone().double().requires_grad_(True) for t in (th, ph, g)]
    beta = torch.softmax(torch.bmm(ref[0].transpose(1, 2), ref[1]), dim=-1)
    o_ref = torch.bmm(ref[2], beta.transpose(1, 2))
    (o_ref * go.double()).sum().backward()
    dv = [t.to(dev()).requires_grad_(True) for t in (th, ph, g)]
    assert TF.nonlocal_attend_ok(4, 16) and not TF.nonlocal_attend_ok(16, 64)
    o = TF.nonlocal_attend(*dv)
    close(o, o_ref, rtol=1e-5, atol=1e-5)
    (o * go.to(dev())).sum().backward()
    for got, want in zip(dv, ref):
        close(got.grad, want.grad, rtol=1e-4, atol=1e-5)


def test_conv_double_backward():
    """R = || d(sum y*gy)/dx ||^2 differentiated w.r.t. w and gy-side input — the GP pattern."""
    from txt2vid_amd import functional as TF
    x = rnd(1, 2, 8, 3, 5, 5)
    w1 = rnd(2, 16, 8, 3, 3, 3) * 0.2
    w2 = rnd(3, 4, 16, 3, 3, 3) * 0.2

    def run(x, w1, w2, conv, relu):
        h = relu(conv(x, w1, None))
        y = conv(h, w2, None)
        gx, = torch.autograd.grad(y.sum(), x, create_graph=True)
        r = (gx ** 2).sum()
        r.backward()
        return r
    xr, w1r, w2r = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    rr = run(xr, w1r, w2r, lambda a, b, c: F.conv3d(a, b, c, padding=1), F.relu)
    xd = x.to(dev()).requires_grad_(True)
    w1d, w2d = torch.nn.Parameter(w1.to(dev())), torch.nn.Parameter(w2.to(dev()))
    rd = run(xd, w1d, w2d, TF.conv, TF.relu)
    close(rd, rr)
    close(w1d.grad, w1r.grad, rtol=1e-3, atol=1e-3)
    close(w2d.grad, w2r.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('shapes', [[(2, 16, 4, 6, 6)], [(4, 64, 2, 16, 16), (2, 64, 4, 8, 8), (1, 64, 1, 5, 3)],
                                    [(3, 8, 1, 1, 1), (2, 8, 2, 2, 2)]])
def test_relu_conv_grouped_first_and_second_order(shapes):
    """ReLU -> conv pairs (layers.py:230-233): fused gather forward, masked-epilogue data gradient (T2V_CONV_MASK_OUT),
    grouped over several tensors that share the weight; gradients of a gradient-penalty-shaped loss vs torch."""
    from txt2vid_amd import functional as TF
    cin, cout = shapes[0][1], 24
    w0, b0 = rnd(5, cout, cin, 3, 3, 3) * 0.2, rnd(6, cout) * 0.1
    xs0 = [rnd(10 + i, *s) for i, s in enumerate(shapes)]

    def run(conv_all, to):
        w, b = to(w0).requires_grad_(True), to(b0).requires_grad_(True)
        xs = [to(x).requires_grad_(True) for x in xs0]
        ys = conv_all(xs, w, b)
        out = sum((y * y).sum() for y in ys)
        gxs = torch.autograd.grad(out, xs, create_graph=True)
        loss = out * 0.01 + sum((g * g).sum() for g in gxs)
        loss.backward()
        return [y.detach() for y in ys], [g.detach() for g in gxs], w.grad, b.grad, [x.grad for x in xs]

    ref = run(lambda xs, w, b: [F.conv3d(F.relu(x), w, b, padding=1) for x in xs], lambda t: t.double())
    if len(shapes) == 1:
        got = run(lambda xs, w, b: [TF.relu_conv(xs[0], w, b)], lambda t: t.to(dev()))
    else:
        got = run(lambda xs, w, b: TF.conv_group(xs, w, b, relu_in=True), lambda t: t.to(dev()))
    for a, r in zip(got[0] + got[1] + [got[2], got[3]] + got[4], ref[0] + ref[1] + [ref[2], ref[3]] + ref[4]):
        close(a, r.float(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('cin,cout,k,shapes', [
    (64, 96, (3, 3, 3), [(2, 64, 4, 8, 8), (1, 64, 2, 16, 16), (3, 64, 1, 4, 4)]),     # 3-tap-row kernel: bias summed on the side
    (64, 64, (1, 3, 3), [(4, 64, 1, 16, 16)]),                                          # 2-D, one member
    (32, 48, (3, 3, 3), [(2, 32, 4, 8, 8), (2, 32, 2, 4, 4)]),                          # Cin < 64: (tap, ci) column-tile kernel with its own bias side sums
    (128, 40, (1, 1, 1), [(2, 128, 4, 8, 8), (5, 128, 1, 1, 1)]),                       # 1x1x1 kernel (per-tap kernel, bias on the side)
    (1, 64, (3, 3, 3), [(2, 1, 4, 8, 8), (3, 1, 2, 16, 16), (2, 1, 1, 5, 3)]),          # the stem's Cin = 1: 27 columns in one tile, ragged member
    (3, 72, (3, 3, 3), [(2, 3, 4, 8, 8)]),                                              # RGB stem, two output-channel tiles x two column tiles
    (64, 32, (3, 3, 3), [(3, 64, 2, 1, 1), (2, 64, 1, 1, 1)]),                          # 3^3 kernel on W = 1 maps: centre-tap workgroups sum the bias
])
def test_wgrad_with_bias_gradient(cin, cout, k, shapes):
    """`t2v_conv_wgrad_grouped_bias`: dW and db of a grouped convolution in one call, stored or accumulated."""
    from txt2vid_amd import functional as TF
    xs = [rnd(60 + i, *s) for i, s in enumerate(shapes)]
    gys = [rnd(70 + i, s[0], cout, *s[2:]) for i, s in enumerate(shapes)]
    w = torch.zeros(cout, cin, *k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    pad = tuple(kk // 2 for kk in k)
    sum((F.conv3d(x.double(), w, b, padding=pad) * g.double()).sum() for x, g in zip(xs, gys)).backward()
    xd, gd = [x.to(dev()) for x in xs], [g.to(dev()) for g in gys]
    dw = torch.full((cout, cin) + k, 7.0, device=dev())
    db = torch.full((cout,), -3.0, device=dev())
    TF.conv_group_wgrad_raw(xd, gd, (cout, cin) + k, False, out=dw, accum=False, dbias=db, accum_bias=False)
    close(dw, w.grad.float(), rtol=2e-4, atol=2e-4)
    close(db, b.grad.float(), rtol=2e-4, atol=2e-4)
    TF.conv_group_wgrad_raw(xd, gd, (cout, cin) + k, False, out=dw, accum=True, dbias=db, accum_bias=True)
    close(dw, 2 * w.grad.float(), rtol=2e-4, atol=2e-4)
    close(db, 2 * b.grad.float(), rtol=2e-4, atol=2e-4)


def test_conv_multi_group_accumulates_data_gradients_in_kernel():
    """`ConvMultiG`: a 3^3 ReLU-conv and a 1^3 conv of the same tensors (a DownBlock's main / skip pair). First-order backward
    without a recorded graph takes the in-kernel accumulation path (T2V_CONV_ACCUM into one buffer per member); the
    create_graph sweep takes the composed path; both against torch, one member without gradient."""
    from txt2vid_amd import functional as TF
    shapes = [(2, 16, 4, 6, 6), (1, 16, 2, 8, 8), (3, 16, 1, 4, 4)]
    w1, b1 = rnd(1, 24, 16, 3, 3, 3) * 0.2, rnd(2, 24) * 0.1
    w2, b2 = rnd(3, 24, 16, 1, 1, 1) * 0.3, rnd(4, 24) * 0.1
    xs0 = [rnd(90 + i, *s_) for i, s_ in enumerate(shapes)]

    def run(fwd, to, second_order):
        ps = [to(t).requires_grad_(True) for t in (w1, b1, w2, b2)]
        xs = [to(x).requires_grad_(i != 2) for i, x in enumerate(xs0)]
        hs, ss = fwd(xs, *ps)
        out = sum(((h + s_) ** 2).sum() for h, s_ in zip(hs, ss))
        if second_order:
            g1 = torch.autograd.grad(out, xs[:2], create_graph=True)
            out = out * 0.01 + sum((g * g).sum() for g in g1)
        out.backward()
        return [h.detach() for h in hs] + [s_.detach() for s_ in ss] + [xs[0].grad, xs[1].grad] + [p_.grad for p_ in ps]

    def ref_fwd(xs, a, b, c, d):
        return [F.conv3d(F.relu(x), a, b, padding=1) for x in xs], [F.conv3d(x, c, d) for x in xs]

    for second in (False, True):
        ref = run(ref_fwd, lambda t: t.double(), second)
        got = run(lambda xs, a, b, c, d: TF.conv_multi_group(xs, [(a, b, True), (c, d, False)]), lambda t: t.to(dev()), second)
        for a_, r_ in zip(got, ref):
            close(a_, r_.float(), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('shapes,cin,cout,k', [
    ([(2, 64, 4, 8, 8)], 64, 96, (3, 3, 3)),
    ([(4, 64, 2, 16, 16), (2, 64, 4, 8, 8), (3, 64, 1, 1, 1)], 64, 64, (3, 3, 3)),        # grouped, one 1x1x1 member
    ([(2, 128, 1, 16, 16)], 128, 40, (1, 3, 3)),                                           # 2-D, Cout not a tile multiple
    ([(8, 32, 8, 32, 32)], 32, 128, (1, 1, 1)),                                            # big M: 128-voxel tiles
    ([(3, 256, 1, 2, 2)], 256, 512, (1, 3, 3)),                                            # tiny M, long K: split-K
])
def test_bf16_compute_convolution(shapes, cin, cout, k):
    """bf16-compute mode (`T2V_CONV_PRECISION=bf16` / `set_conv_precision('bf16')`, BASELINE configs 2-4): forward (+bias, fused
    input ReLU) and data gradient (+ masked epilogue) equal the EXACT convolution of the bf16-rounded operands (fp32
    accumulation) and stay within bf16 rounding of the fp32 result."""
    from txt2vid_amd import functional as TF
    xs = [rnd(300 + i, *sh) for i, sh in enumerate(shapes)]
    w, b = rnd(310, cout, cin, *k) * 0.1, rnd(311, cout) * 0.1
    pad = tuple(kk // 2 for kk in k)
    r16 = lambda t: t.bfloat16().double()
    old = TF.set_conv_precision('bf16')
    try:
        wd = torch.nn.Parameter(w.to(dev()))
        ys = TF.conv_group_raw([x.to(dev()) for x in xs], wd, b.to(dev()), True, 0)
        gys = [rnd(320 + i, *y.shape) for i, y in enumerate(ys)]
        masks = [rnd(330 + i, *x.shape) for i, x in enumerate(xs)]
        gxs = TF.conv_group_raw([g.to(dev()) for g in gys], wd, None, False, 1, masks=[m.to(dev()) for m in masks])
    finally:
        TF.set_conv_precision(old)
    for x, y, g, m, gx in zip(xs, ys, gys, masks, gxs):
        exact = F.conv3d(r16(F.relu(x)), r16(w), b.double(), padding=pad)
        close(y, exact.float(), rtol=2e-5, atol=2e-5)
        full = F.conv3d(F.relu(x).double(), w.double(), b.double(), padding=pad)
        assert float((y.cpu().double() - full).abs().max()) <= 2e-2 * float(full.abs().max())
        rg = r16 if cout % 32 == 0 else (lambda t: t.double())       # K = Cout of the data gradient: not a multiple of 32 -> fp32 kernel
        exact_g = F.conv_transpose3d(rg(g), rg(w), padding=pad) * (m > 0).double()
        close(gx, exact_g.float(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('shapes,cin,cout,k', [
    ([(2, 64, 4, 8, 8), (1, 64, 2, 16, 16)], 64, 96, (3, 3, 3)),
    ([(4, 128, 1, 16, 16)], 128, 64, (1, 3, 3)),
    ([(2, 128, 4, 8, 8), (5, 128, 1, 1, 1)], 128, 40, (1, 1, 1)),            # per-tap kernel (1x1x1)
    ([(3, 64, 2, 1, 1)], 64, 32, (3, 3, 3)),                                 # per-tap kernel, 3^3 on W = 1 maps
])
def test_bf16_compute_weight_gradient(shapes, cin, cout, k):
    """bf16-compute mode, 3-tap-row weight gradient (+ the bias side-sum, which stays fp32): equals the EXACT weight gradient
    of the bf16-rounded operands (fp32 accumulation), with the fused input ReLU."""
    from txt2vid_amd import functional as TF
    xs = [rnd(400 + i, *sh) for i, sh in enumerate(shapes)]
    gys = [rnd(410 + i, sh[0], cout, *sh[2:]) for i, sh in enumerate(shapes)]
    pad = tuple(kk // 2 for kk in k)
    r16 = lambda t: t.bfloat16().double()
    w = torch.zeros(cout, cin, *k, dtype=torch.float64, requires_grad=True)
    sum((F.conv3d(r16(F.relu(x)), w, None, padding=pad) * r16(g)).sum() for x, g in zip(xs, gys)).backward()
    db_ref = sum(g.double().sum(dim=(0, 2, 3, 4)) for g in gys)
    old = TF.set_conv_precision('bf16')
    try:
        dw = torch.empty((cout, cin) + k, device=dev())
        db = torch.empty((cout,), device=dev())
        TF.conv_group_wgrad_raw([x.to(dev()) for x in xs], [g.to(dev()) for g in gys], (cout, cin) + k, True, out=dw, dbias=db)
    finally:
        TF.set_conv_precision(old)
    close(dw, w.grad.float(), rtol=5e-5, atol=5e-5)
    close(db, db_ref.float(), rtol=1e-5, atol=1e-5)


def test_avgpool():
    from txt2vid_amd import functional as TF
    for shape, k, s, p in (((2, 3, 4, 6, 6), (1, 2, 2), (2, 2, 2), (0, 0, 0)),
                           ((2, 3, 3, 5, 7), (2, 2, 2), (2, 2, 2), (1, 1, 1)),
                           ((1, 2, 1, 4, 1), (1, 2, 1), (1, 2, 1), (0, 0, 0))):
        x = rnd(1, *shape)
        xr = x.clone().requires_grad_(True)
        yr = F.avg_pool3d(xr, k, s, p)
        gy = rnd(2, *yr.shape)
        (yr * gy).sum().backward()
        xd = x.to(dev()).requires_grad_(True)
        yd = TF.avg_pool3d(xd, k, s, p)
        close(yd, yr)
        (yd * gy.to(dev())).sum().backward()
        close(xd.grad, xr.grad)


def test_avgpool_group_matches_torch_first_and_second_order():
    """Multi-tensor pooling (`t2v_avgpool3d_multi` / `_bwd_multi`): pool(x + x2) + add over four differently shaped
    members, incl. the stem's kernel (1,2,2) / stride 2 and an identity member; gradients of a GP-shaped loss."""
    from txt2vid_amd import functional as TF
    shapes = [(2, 3, 4, 8, 8), (1, 3, 2, 6, 4), (3, 3, 1, 4, 4), (2, 3, 1, 1, 1)]
    cfgs = [((2, 2, 2), (2, 2, 2), (0, 0, 0)), ((1, 2, 2), (2, 2, 2), (0, 0, 0)), ((1, 2, 2), (1, 2, 2), (0, 0, 0)),
            ((1, 1, 1), (1, 1, 1), (0, 0, 0))]
    xs0 = [rnd(30 + i, *s) for i, s in enumerate(shapes)]
    x2s0 = [rnd(40 + i, *s) for i, s in enumerate(shapes)]

    def run(pool, to):
        xs = [to(x).requires_grad_(True) for x in xs0]
        x2s = [to(x).requires_grad_(True) for x in x2s0]
        ys0 = pool(xs, x2s, None)
        adds = [to(rnd(50 + i, *y.shape)).requires_grad_(True) for i, y in enumerate(ys0)]
        ys = pool(xs, x2s, adds)
        out = sum((y ** 3).sum() for y in ys)
        g1 = torch.autograd.grad(out, xs + adds, create_graph=True)
        loss = out + sum((g * g).sum() for g in g1)
        loss.backward()
        return [y.detach() for y in ys] + [t.grad for t in xs + x2s + adds]

    def torch_pool(xs, x2s, adds):
        ys = [F.avg_pool3d(x + x2, k, s, p) for x, x2, (k, s, p) in zip(xs, x2s, cfgs)]
        return ys if adds is None else [y + a for y, a in zip(ys, adds)]

    ref = run(torch_pool, lambda t: t.double())
    got = run(lambda xs, x2s, adds: TF.avg_pool3d_group(xs, cfgs, x2s=x2s, adds=adds), lambda t: t.to(dev()))
    for a, r in zip(got, ref):
        close(a, r.float(), rtol=1e-4, atol=1e-4)


def test_maxpool_softmax_bmm_double_backward():
    from txt2vid_amd import functional as TF
    th, ph, gg = rnd(1, 2, 4, 16), rnd(2, 2, 4, 2, 4, 4), rnd(3, 2, 8, 2, 4, 4)

    def run(th, ph, gv, mp, bmm, sm):
        p = mp(ph).reshape(2, 4, -1)
        g = mp(gv).reshape(2, 8, -1)
        beta = sm(bmm(th, p, True, False))                 # [2,16,8]
        o = bmm(g, beta, False, True)                      # [2,8,16]
        go, = torch.autograd.grad((o ** 2).sum(), th, create_graph=True)
        r = (go ** 2).sum() + o.sum()
        r.backward()
        return o, r

    def ref_bmm(a, b, ta, tb):
        return torch.bmm(a.transpose(1, 2) if ta else a, b.transpose(1, 2) if tb else b)
    a = [t.clone().requires_grad_(True) for t in (th, ph, gg)]
    o_r, r_r = run(a[0], a[1], a[2], lambda t: F.max_pool3d(t, [1, 2, 2]), ref_bmm, lambda t: F.softmax(t, -1))
    d = [t.to(dev()).requires_grad_(True) for t in (th, ph, gg)]
    o_d, r_d = run(d[0], d[1], d[2], TF.max_pool2x2, TF.bmm, TF.softmax_lastdim)
    close(o_d, o_r)
    close(r_d, r_r)
    for td, tr in zip(d, a):
        close(td.grad, tr.grad, rtol=1e-3, atol=1e-3)


def test_nonlocal_block_grouped_matches_per_level_first_and_second_order():
    """`layers.nonlocal_levels` (every op one multi-job launch over the levels, `t2v_multi`) against the per-tensor
    non-local block on three differently shaped members, one of which takes no part in the backward; gradients of a
    gradient-penalty-shaped loss (double backward through max-pool, both batched GEMMs, softmax and gamma*o + x)."""
    from txt2vid_amd.models.layers import Attention3d, nonlocal_levels
    torch.manual_seed(3)
    att = Attention3d(32).to(dev())
    with torch.no_grad():
        att.gamma.fill_(0.7)
        for p_ in att.parameters():
            if p_.dim() > 1:
                p_.mul_(3.0)
    shapes = [(2, 32, 4, 4, 4), (1, 32, 2, 8, 8), (3, 32, 1, 2, 2)]
    xs0 = [rnd(80 + i, *sh) for i, sh in enumerate(shapes)]

    def run(fwd):
        for p_ in att.parameters():
            p_.grad = None
        xs = [x.to(dev()).requires_grad_(i != 2) for i, x in enumerate(xs0)]
        ys = fwd(xs)
        out = sum((y ** 2).sum() for y in ys[:2]) + ys[2].sum() * 0.0
        g1 = torch.autograd.grad(out, xs[:2], create_graph=True)
        loss = out * 0.1 + sum((g * g).sum() for g in g1)
        loss.backward()
        return [y.detach() for y in ys] + [g.detach() for g in g1] + [xs[0].grad, xs[1].grad] + [p_.grad.clone() for p_ in att.parameters()]

    ref = run(lambda xs: [att(x) for x in xs])
    got = run(lambda xs: nonlocal_levels(att, xs))
    for a, r in zip(got, ref):
        close(a, r, rtol=2e-4, atol=2e-5)


def test_scale_add_rowsum_double_backward():
    from txt2vid_amd import functional as TF
    gam, o, x = torch.tensor(0.7), rnd(1, 2, 3, 2, 2, 2), rnd(2, 2, 3, 2, 2, 2)

    def run(gam, o, x, scale_add, ssum):
        y = scale_add(gam, o * o, x)
        f = ssum(y * y)
        g, = torch.autograd.grad(f.sum(), o, create_graph=True)
        r = (g ** 2).sum()
        r.backward()
        return f, r
    a = [t.clone().requires_grad_(True) for t in (gam, o, x)]
    f_r, r_r = run(a[0], a[1], a[2], lambda g, o, x: g * o + x, lambda t: t.sum([2, 3, 4]))
    d = [t.to(dev()).requires_grad_(True) for t in (gam, o, x)]
    f_d, r_d = run(d[0], d[1], d[2], TF.scale_add, TF.sum_spatial)
    close(f_d, f_r)
    close(r_d, r_r)
    for td, tr in zip(d, a):
        close(td.grad, tr.grad, rtol=1e-3, atol=1e-3)


def test_batchnorm_tanh_upsample():
    from txt2vid_amd import functional as TF
    x = rnd(1, 4, 6, 5, 5)
    gamma, beta = 1 + 0.1 * rnd(2, 6), 0.1 * rnd(3, 6)
    rm, rv = 0.1 * rnd(4, 6), 1 + 0.1 * torch.rand(6)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    yr = torch.tanh(F.interpolate(F.relu(F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)), scale_factor=2))
    gy = rnd(5, *yr.shape)
    (yr * gy).sum().backward()
    xd = x.to(dev()).requires_grad_(True)
    gd, bd = torch.nn.Parameter(gamma.to(dev())), torch.nn.Parameter(beta.to(dev()))
    rmd, rvd = rm.to(dev()), rv.to(dev())
    yd = TF.tanh(TF.upsample2x(TF.batch_norm_act(xd, gd, bd, rmd, rvd, True, 0.1, 1e-5, True)))
    close(yd, yr)
    (yd * gy.to(dev())).sum().backward()
    close(xd.grad, xr.grad, rtol=1e-3, atol=1e-4)
    close(gd.grad, gr.grad, rtol=1e-3, atol=1e-4)
    close(bd.grad, br.grad, rtol=1e-3, atol=1e-4)
    close(rmd, rmr)
    close(rvd, rvr)
    # eval mode
    ye = TF.batch_norm_act(x.to(dev()), gd, bd, rmd, rvd, False, 0.1, 1e-5, False)
    close(ye, F.batch_norm(x, rmr, rvr, gamma, beta, False, 0.1, 1e-5))


@pytest.mark.parametrize('members,cin,cout', [
    ([(2, 4, 16, 16), (1, 5, 8, 8), (3, 2, 4, 4)], 64, 64),         # even, odd and two-frame members (64-voxel strip3 tiles)
    ([(8, 16, 32, 32)], 64, 96),                                     # M_out = 65536: 256-voxel strip3 tiles, two channel tiles
])
def test_conv_even_frames_group(members, cin, cout):
    """`t2v_conv_group.dstride = 2` (ConvEvenFramesG): ReLU -> conv3^3 evaluated on the EVEN output frames only (what the stem's
    AvgPool3d((1,2,2), stride 2) keeps): the even frames of the full launch, and the same gradients — first
    order and through a recorded backward — as torch's conv followed by [:, :, ::2]."""
    from txt2vid_amd import functional as TF
    xs_h = [rnd(90 + i, n, cin, d, h, w) for i, (n, d, h, w) in enumerate(members)]
    w_h, b_h = rnd(7, cout, cin, 3, 3, 3) * 0.05, rnd(8, cout)
    xs = [t.to(dev()).requires_grad_(True) for t in xs_h]
    w, b = torch.nn.Parameter(w_h.to(dev())), torch.nn.Parameter(b_h.to(dev()))
    assert TF.even_frames_ok(xs, w)
    ys = TF.conv_even_frames_group(xs, w, b, relu_in=True)
    full = TF.conv_group([t.detach() for t in xs], w.detach(), b.detach(), relu_in=True)
    for y, f, (n, d, h, wd) in zip(ys, full, members):
        assert y.shape == (n, cout, (d + 1) // 2, h, wd)
        close(y.detach(), f[:, :, ::2], rtol=1e-5, atol=1e-5)       # (same GEMM; the full launch may split K, so not bit for bit)
    gys = [rnd(50 + i, *y.shape) for i, y in enumerate(ys)]

    def second_order(outs, leaves, to):
        f = sum((o * g.to(to)).sum() for o, g in zip(outs, gys))
        g1 = torch.autograd.grad(f, leaves, create_graph=True)
        pen = sum((g * g).sum() for g in g1[:len(outs)])                     # (penalty on the data gradients, like the GP)
        g2 = torch.autograd.grad(pen, leaves[len(outs):], allow_unused=True)
        return [g.detach() for g in g1], g2

    got1, got2 = second_order(ys, xs + [w, b], dev())
    xr = [t.clone().requires_grad_(True) for t in xs_h]
    wr, br = w_h.clone().requires_grad_(True), b_h.clone().requires_grad_(True)
    yr = [F.conv3d(F.relu(x), wr, br, padding=1)[:, :, ::2] for x in xr]
    ref1, ref2 = second_order(yr, xr + [wr, br], 'cpu')
    for a, r in zip(got1, ref1):
        close(a, r, rtol=2e-4, atol=2e-4)                                   # (same bounds as every other fp32 conv case)
    close(got2[0], ref2[0], rtol=2e-4, atol=2e-4)                           # d penalty / d w through the recorded backward
    # first order WITHOUT a recorded graph: the data gradient comes from two strided-output launches on the even-frame gradients
    ys = TF.conv_even_frames_group(xs, w, b, relu_in=True)
    plain = torch.autograd.grad(sum((o * g.to(dev())).sum() for o, g in zip(ys, gys)), xs + [w, b])
    for a, r in zip(plain, ref1):
        close(a, r, rtol=2e-4, atol=2e-4)
    assert got2[1] is None or float(got2[1].abs().max()) == 0.0            # the bias does not enter the data gradient


@pytest.mark.parametrize('members,cin,cout', [
    ([(2, 4, 16, 16), (1, 5, 8, 8), (3, 2, 4, 4)], 64, 64),         # ragged members, odd frame count, non-power-of-two decode
    ([(4, 16, 16, 16), (2, 8, 32, 32)], 64, 96),                     # power-of-two extents (shift decode), two channel tiles
])
def test_weight_gradient_from_even_frame_gradients(members, cin, cout):
    """`t2v_conv_wgrad_grouped[_bias]` with dstride = 2: dL/dy lives on the even frames only. dW / db against torch on the CPU:
    the weight / bias gradient of `conv3d(relu(x), w)[:, :, ::2]` (fp64 accumulation of the per-member fp32 references, like the
    other weight-gradient cases), and bit-level agreement is NOT assumed with the library's own full-frame call."""
    from txt2vid_amd import functional as TF
    xs_h = [rnd(30 + i, n, cin, d, h, w) for i, (n, d, h, w) in enumerate(members)]
    ge_h = [rnd(40 + i, n, cout, (d + 1) // 2, h, w) for i, (n, d, h, w) in enumerate(members)]
    shape = (cout, cin, 3, 3, 3)
    dw_ref, db_ref = torch.zeros(shape, dtype=torch.float64), torch.zeros(cout, dtype=torch.float64)
    for x, g in zip(xs_h, ge_h):
        wr, br = torch.zeros(shape, requires_grad=True), torch.zeros(cout, requires_grad=True)
        (F.conv3d(F.relu(x), wr, br, padding=1)[:, :, ::2] * g).sum().backward()
        dw_ref += wr.grad.double()
        db_ref += br.grad.double()
    xs, ge = [t.to(dev()) for t in xs_h], [t.to(dev()) for t in ge_h]
    dw, db = torch.full(shape, 3.0, device=dev()), torch.full((cout,), -2.0, device=dev())
    TF.conv_group_wgrad_raw(xs, ge, shape, True, out=dw, dbias=db, even_frames=True)
    close(dw, dw_ref, rtol=2e-4, atol=2e-4)
    close(db, db_ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('case', BF16_CASES, ids=[c[0] for c in BF16_CASES])
def test_bf16_cases(case):
    """bf16-compute mode at the sizes (and in the frame-strided forms) the benchmark's bf16 iteration launches — every compiled
    `conv_igemm_bf16*` / bf16 weight-gradient instantiation (tests/test_conv_plan.py checks that claim on the host): forward
    (+ bias, fused input ReLU; `even`: on the even output frames, `dstride = 2`), data gradient (ReLU adjoint in the epilogue;
    `even`: the two `ydstride = 2` launches from even-frame dL/dy) and weight + bias gradient (`even`: from even-frame dL/dy)
    against the convolution of the bf16-ROUNDED operands on the CPU. Products of bf16 values are exact in fp32, so the fp32
    reference differs from the kernels' fp32 accumulation by summation order only: same 1e-4 / 2e-4 bounds as the fp32 cases
    (a wrong halo row, tap list or frame offset is an O(1) error)."""
    from txt2vid_amd import functional as TF
    name, cin, cout, k, members, relu_in, even = case
    r16 = lambda t: t.bfloat16().float()
    w = rnd(2, cout, cin, *k) * (1.0 / np.sqrt(cin * float(np.prod(k))))
    b = rnd(3, cout) * 0.1
    xs = [rnd(10 + i, n, cin, d, h, wd) for i, (n, d, h, wd) in enumerate(members)]
    gys = [rnd(40 + i, n, cout, (d + 1) // 2 if even else d, h, wd) for i, (n, d, h, wd) in enumerate(members)]
    pad = tuple(kk // 2 for kk in k)
    old = TF.set_conv_precision('bf16')
    try:
        wd_, bd_ = torch.nn.Parameter(w.to(dev())), b.to(dev())
        xd, gyd = [x.to(dev()) for x in xs], [g.to(dev()) for g in gys]
        dbias = torch.empty(cout, device=dev())
        if even:
            assert TF.even_frames_ok(xd, wd_)
            ys = TF.conv_group_raw(xd, wd_, bd_, relu_in, 0, even_frames=True)
            gxs = TF._dgrad_even_frames_raw(gyd, wd_, xd)
            assert gxs is not None
        else:
            ys = TF.conv_group_raw(xd, wd_, bd_, relu_in, 0)
            gxs = TF.conv_group_raw(gyd, wd_, None, False, 1, masks=xd if relu_in else None)
        dw = TF.conv_group_wgrad_raw(xd, gyd, tuple(w.shape), relu_in, dbias=dbias, even_frames=even)
        torch.cuda.synchronize()
    finally:
        TF.set_conv_precision(old)
    dw_ref = torch.zeros(w.shape, dtype=torch.float64)
    db_ref = torch.zeros(cout, dtype=torch.float64)
    for i, (x, gy) in enumerate(zip(xs, gys)):
        xr, wr = r16(x).requires_grad_(True), r16(w).requires_grad_(True)
        yr = F.conv3d(F.relu(xr) if relu_in else xr, wr, b, padding=pad)
        if even:
            yr = yr[:, :, ::2]
        close(ys[i], yr)
        (yr * r16(gy)).sum().backward()
        close(gxs[i], xr.grad)
        dw_ref += wr.grad.double()
        db_ref += gy.double().sum(dim=(0, 2, 3, 4))                  # (the bias side-sum stays fp32: unrounded dL/dy)
    close(dw, dw_ref, rtol=2e-4, atol=2e-4)
    close(dbias, db_ref, rtol=2e-4, atol=2e-4)


def test_cat_features_group_first_and_second_order():
    """`T2V_MJ_CATCOLS / SLICECOLS / EMBEDCOLS`: torch.cat((features, cond), 1) of every level in one launch, closed under
    differentiation (the conditional heads take part in the gradient penalty's double backward)."""
    from txt2vid_amd import functional as TF
    shapes = [(8, 12, 5), (4, 12, 5), (1, 12, 5)]
    a_h = [rnd(60 + i, r, na) for i, (r, na, nb) in enumerate(shapes)]
    b_h = [rnd(70 + i, r, nb) for i, (r, na, nb) in enumerate(shapes)]
    w_h = [rnd(80 + i, r, na + nb) for i, (r, na, nb) in enumerate(shapes)]

    def run(cat, to):
        a = [t.to(to).requires_grad_(True) for t in a_h]
        b = [t.to(to).requires_grad_(i != 1) for i, t in enumerate(b_h)]
        outs = cat(a, b)
        f = sum((w.to(to) * o * o * o).sum() for w, o in zip(w_h, outs))
        live = a + [t for t in b if t.requires_grad]
        g = torch.autograd.grad(f, live, create_graph=True)
        gg = torch.autograd.grad(sum((x * x).sum() for x in g), live)
        return [o.detach() for o in outs], [x.detach() for x in g], gg

    ref = run(lambda a, b: [torch.cat((x, y), 1) for x, y in zip(a, b)], 'cpu')
    got = run(TF.cat_features_group, dev())
    for r, g in zip(ref[0], got[0]):
        assert torch.equal(g.cpu(), r)
    for r, g in zip(ref[1] + list(ref[2]), got[1] + list(got[2])):
        close(g, r, rtol=1e-4, atol=1e-4)


def test_rsgan_mean_over_levels_is_the_per_level_loss_and_mean_bit_for_bit():
    """`t2v_rsgan_mean_multi[_bwd]`: RSGANLoss of every pyramid level and their mean (cond_gan.py:121-154) in one launch each
    way; the same numbers, bit for bit, as `rsgan` per level + `scalar_mean`, and torch's BCEWithLogits within rounding."""
    from txt2vid_amd import functional as TF
    ns = [64, 32, 16, 7]
    a = [rnd(40 + i, n, 1).to(dev()).requires_grad_(True) for i, n in enumerate(ns)]
    b = [rnd(50 + i, n, 1).to(dev()).requires_grad_(i != 2) for i, n in enumerate(ns)]
    one = TF.rsgan_mean_levels(a, b)
    ref = TF.scalar_mean([TF.rsgan(x, y) for x, y in zip(a, b)])
    assert torch.equal(one, ref)
    live = a + [t for t in b if t.requires_grad]
    g1 = torch.autograd.grad(one * 1.7, live)
    g2 = torch.autograd.grad(ref * 1.7, live)
    assert all(torch.equal(u, v) for u, v in zip(g1, g2))
    tref = sum(F.binary_cross_entropy_with_logits(x.detach().cpu() - y.detach().cpu(), torch.ones(n, 1)) for x, y, n in zip(a, b, ns)) / len(ns)
    close(one, tref, rtol=1e-6, atol=1e-6)


def test_cat_lerp_group_builds_the_d_step_inputs_in_one_launch():
    """`T2V_MJ_CATLERP`: torch.cat((real, fake)) and alpha*real + (1-alpha)*fake (losses.py:146) for every pyramid level."""
    from txt2vid_amd import functional as TF
    shapes = [(4, 1, 8, 8, 8), (2, 1, 4, 16, 16), (1, 3, 2, 5, 7)]
    reals = [rnd(20 + i, *s) for i, s in enumerate(shapes)]
    fakes = [rnd(30 + i, *s) for i, s in enumerate(shapes)]
    alphas = [torch.rand(s[0]) for s in shapes]
    rfs, xhs = TF.cat_lerp_group([t.to(dev()) for t in reals], [t.to(dev()) for t in fakes], [a.to(dev()) for a in alphas])
    for r, f, a, rf, xh in zip(reals, fakes, alphas, rfs, xhs):
        assert torch.equal(rf.cpu(), torch.cat((r, f)))
        av = a.view(-1, *([1] * (r.dim() - 1)))
        close(xh, av * r + (1 - av) * f, rtol=1e-6, atol=1e-6)
    rfs, xhs = TF.cat_lerp_group([t.to(dev()) for t in reals], [t.to(dev()) for t in fakes])
    assert xhs is None and all(torch.equal(rf.cpu(), torch.cat((r, f))) for r, f, rf in zip(reals, fakes, rfs))


def test_fork_group_sums_gradients_in_one_launch_first_and_second_order():
    """`fork_group`: two aliases per tensor whose gradients are summed by ONE grouped launch (`T2V_MJ_ADD`) instead of one
    autograd-engine add per tensor; members that need no gradient pass through; closed under double backward."""
    from txt2vid_amd import functional as TF
    xs = [rnd(5 + i, *s).to(dev()).requires_grad_(i != 1) for i, s in enumerate([(2, 3, 4), (3, 5), (1, 7, 2, 2)])]
    ws = [rnd(15 + i, *x.shape).to(dev()) for i, x in enumerate(xs)]
    a, b = TF.fork_group(xs)
    assert a[1] is xs[1] and b[1] is xs[1]                                   # no gradient wanted: untouched
    assert all(torch.equal(t, x) and torch.equal(u, x) for t, u, x in zip(a, b, xs))
    # f = sum w * a^2 * b  (= w x^3): df/dx = 3 w x^2 through the two aliases, d/dx of sum (df/dx)^2 = 36 w^2 x^3
    f = sum((w * t * t * u).sum() for w, t, u in zip(ws, a, b))
    live = [xs[0], xs[2]]
    g = torch.autograd.grad(f, live, create_graph=True)
    for gi, x, w in zip(g, live, [ws[0], ws[2]]):
        close(gi, 3 * w * x.detach() ** 2, rtol=1e-5, atol=1e-5)
    gg = torch.autograd.grad(sum((gi * gi).sum() for gi in g), live)
    for ggi, x, w in zip(gg, live, [ws[0], ws[2]]):
        close(ggi, 36 * w * w * x.detach() ** 3, rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        a, b = TF.fork_group(xs)
        assert all(t is x and u is x for t, u, x in zip(a, b, xs))


@pytest.mark.parametrize('cin,cout', [(16, 16), (32, 16)])
def test_upblock_identity_path_on_the_small_map(cin, cout):
    """UpBlock (layers.py:152-195): the identity path Up [-> conv1x1] is evaluated as conv1x1 on the small map and one
    `t2v_upsample2x_add` launch; output identical to the reference order of operations (torch modules), gradients of x and of
    every parameter within rounding."""
    from txt2vid_amd.models.layers import UpBlock
    torch.manual_seed(3)
    blk = UpBlock(in_channels=cin, out_channels=cout)
    x = rnd(11, 3, cin, 6, 6)
    m, idm = blk.main.inner_module, blk.main.identity_map
    xr = x.clone().requires_grad_(True)
    h = F.relu(F.batch_norm(xr, None, None, m[0].weight, m[0].bias, True, 0.1, 1e-5))
    h = F.conv2d(F.interpolate(h, scale_factor=2), m[3].weight, m[3].bias, padding=1)
    h = F.relu(F.batch_norm(h, None, None, m[4].weight, m[4].bias, True, 0.1, 1e-5))
    h = F.conv2d(h, m[6].weight, m[6].bias, padding=1)
    s = F.interpolate(xr, scale_factor=2)
    if cin != cout:
        s = F.conv2d(s, idm[1].weight, idm[1].bias)
    yr = s + h
    gy = rnd(12, *yr.shape)
    params = list(blk.parameters())
    gref = torch.autograd.grad((yr * gy).sum(), [xr] + params)
    blk = blk.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    yd = blk(xd)
    close(yd, yr.detach(), rtol=1e-4, atol=1e-4)
    gd = torch.autograd.grad((yd * gy.to(dev())).sum(), [xd] + list(blk.parameters()))
    for a, r in zip(gd, gref):
        close(a, r, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize('hw,C,B', [(1, 16, 3), (2, 16, 3), (1, 128, 5), (1, 256, 33), (1, 1024, 32)])
def test_conv_lstm(hw, C, B):
    """conv_lstm.py:75-97 vs the oracle. On 1x1 maps with B <= 32 and C % 64 == 0 a step is ONE launch (GEMM + gates fused on
    the unit-major weight copy: C = 128 the generic instantiation with a ragged row tile, [32,1024,1,1] the generator's own
    case); B = 33 takes the wave-per-strip GEMM + slab-summing gate kernels (a second, ragged MFMA row tile)."""
    from txt2vid_amd import functional as TF
    from oracle import tganv2_oracle as O
    steps = 5
    shapes = {}
    for gate in 'ifco':
        shapes['Wx%s.weight' % gate] = (C, C, 3, 3)
        shapes['Wx%s.bias' % gate] = (C,)
        shapes['Wh%s.weight' % gate] = (C, C, 3, 3)
    P = O.recipe_state(shapes)
    for v in P.values():
        v.mul_(3.0 if C == 16 else 3.0 * (16.0 / C) ** 0.5).requires_grad_(True)
    x = rnd(1, B, C, hw, hw)
    xr = x.clone().requires_grad_(True)
    yr = torch.stack(O.conv_lstm(P, '', xr, steps))
    gy = rnd(2, *yr.shape)
    (yr * gy).sum().backward()
    Pd = {k: torch.nn.Parameter(v.detach().to(dev())) for k, v in P.items()}
    xd = x.to(dev()).requires_grad_(True)
    yd = TF.conv_lstm(xd, steps, [Pd['Wx%s.weight' % g] for g in 'ifco'], [Pd['Wx%s.bias' % g] for g in 'ifco'],
                      [Pd['Wh%s.weight' % g] for g in 'ifco'])
    close(yd, yr)
    (yd * gy.to(dev())).sum().backward()
    close(xd.grad, xr.grad, rtol=1e-3, atol=1e-4)
    for k in P:
        close(Pd[k].grad, P[k].grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('shape', [(70, 40, 3, 3, 3), (64, 96, 1, 3, 3), (33, 20, 1, 1, 1), (128, 64, 3, 3, 3), (8, 5, 3, 3, 3)])
def test_repack_params_matches_a_fresh_pack(shape):
    """`repack_params` (the ONE multi-tensor launch after an optimiser step, functional.py) against `t2v_pack_weight[_bf16]` of the same
    values: forward and mirrored data-gradient layouts, fp32 and bf16, 3^k and 1x1x1 kernels, ragged channel tiles, a member with
    D == 1 (9 live taps of 27)."""
    from txt2vid_amd import functional as TF
    cout, cin, kd, kh, kw = shape
    torch.manual_seed(3)
    w = torch.nn.Parameter(torch.randn(shape).to(dev()))
    geoms = [TF.conv_geom(2, cin, 4, 6, 6, cout, kd, kh, kw)]
    if kd == 3:
        geoms.append(TF.conv_geom(2, cin, 1, 6, 6, cout, kd, kh, kw))               # D == 1: only the centre plane of taps is live
    packed = []
    for g in geoms:
        ts = TF._tapset(g.T, g.mask)
        for mode in (0, 1):
            packed.append((TF.packed_weight(w, g, mode), lambda dst, g=g, mode=mode: TF.check(TF.lib().t2v_pack_weight(
                TF._p(w), TF._p(dst), cout, cin, g.T, g.taps_c, len(g.taps), mode, TF._stream()), 'pack')))
            packed.append((TF.packed_weight_bf16(w, ts, mode), lambda dst, ts=ts, mode=mode: TF.check(TF.lib().t2v_pack_weight_bf16(
                TF._p(w), TF._p(dst), cout, cin, ts.T, ts.taps_c, len(ts.taps), mode, TF._stream()), 'pack16')))
    with torch.no_grad():
        w.copy_(torch.randn(shape).to(dev()))                # new values behind the caches' back ...
    ptrs = [wp.data_ptr() for wp, _ in packed]
    TF.repack_params([w])                                    # ... refreshed in place by the multi-tensor launch
    torch.cuda.synchronize()
    for (wp, fresh), ptr in zip(packed, ptrs):
        assert wp.data_ptr() == ptr
        ref = torch.empty_like(wp)
        fresh(ref)
        torch.cuda.synchronize()
        assert torch.equal(wp.view(torch.int16 if wp.dtype == torch.bfloat16 else torch.int32),
                           ref.view(torch.int16 if wp.dtype == torch.bfloat16 else torch.int32)), (shape, wp.dtype)


def test_losses_and_gp_helpers():
    from txt2vid_amd import functional as TF
    a, b = rnd(1, 6, 1), rnd(2, 6, 1)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    lr = F.binary_cross_entropy_with_logits(ar - br, torch.ones_like(ar))
    (lr * 1.7).backward()
    ad, bd = a.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    ld = TF.rsgan(ad, bd)
    close(ld, lr)
    (ld * 1.7).backward()
    close(ad.grad, ar.grad)
    close(bd.grad, br.grad)
    al, xr_, xf_ = torch.rand(3), rnd(3, 3, 2, 4, 4, 4), rnd(4, 3, 2, 4, 4, 4)
    xh = TF.lerp_rows(al.to(dev()), xr_.to(dev()), xf_.to(dev()))
    a5 = al.view(3, 1, 1, 1, 1)
    close(xh, a5 * xr_ + (1 - a5) * xf_)
    g = rnd(5, 3, 40)
    gr_ = g.clone().requires_grad_(True)
    (gr_.norm(2, dim=1) ** 2).sum().backward()
    gd_ = g.to(dev()).requires_grad_(True)
    sq = TF.row_sqnorm(gd_)
    close(sq, g.norm(2, dim=1) ** 2)
    sq.sum().backward()
    close(gd_.grad, gr_.grad)


@pytest.mark.parametrize('name', ['vanilla', 'hinge', 'hinge3', 'wasserstein', 'rasgan', 'ralsgan'])
def test_loss_zoo_matches_reference_goldens_and_oracle(name):
    """gan/losses.py:19-133 through `t2v_gan_loss(_bwd)`: the committed reference outputs on [12,1] logits, then the
    oracle on ragged / large heads (n = 1, 255, 257, 4099) with a non-unit upstream gradient."""
    import os
    from oracle import tganv2_oracle as O
    from txt2vid_amd.gan import losses as L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'losses.npz'))
    obj = {'vanilla': L.VanillaGanLoss(), 'hinge': L.HingeGanLoss(), 'hinge3': L.HingeGanLoss(margin=3.0),
           'wasserstein': L.WassersteinGanLoss(), 'rasgan': L.RaSGANLoss(), 'ralsgan': L.RaLSGANLoss()}[name]
    kind, margin = ('hinge', 3.0) if name == 'hinge3' else (name, 2.0)
    for side, fn in ((0, obj.discrim_loss), (1, obj.gen_loss)):
        r = torch.tensor(g['real']).to(dev()).requires_grad_(True)
        f = torch.tensor(g['fake']).to(dev()).requires_grad_(True)
        loss = fn(fake=f, real=r)
        loss.backward()
        zero = torch.zeros(12, 1)
        close(loss, torch.tensor(g['%s.%d.loss' % (name, side)]), rtol=1e-5, atol=1e-6)
        close(r.grad if r.grad is not None else zero, torch.tensor(g['%s.%d.g_real' % (name, side)]), rtol=1e-5, atol=1e-6)
        close(f.grad if f.grad is not None else zero, torch.tensor(g['%s.%d.g_fake' % (name, side)]), rtol=1e-5, atol=1e-6)
        for n in (1, 255, 257, 4099):
            r0, f0 = rnd(10 + n, n, 1) * 3, rnd(20 + n, n, 1) * 3 - 0.5
            ro, fo = r0.clone().requires_grad_(True), f0.clone().requires_grad_(True)
            lo = O.zoo_loss(kind, side, fo, ro, margin=margin)
            (lo * -2.5).backward()
            rd, fd = r0.to(dev()).requires_grad_(True), f0.to(dev()).requires_grad_(True)
            ld = fn(fake=fd, real=rd)
            (ld * -2.5).backward()
            close(ld, lo, rtol=2e-5, atol=2e-6)
            close(rd.grad if rd.grad is not None else torch.zeros(n, 1), ro.grad if ro.grad is not None else torch.zeros(n, 1),
                  rtol=1e-4, atol=1e-7)
            close(fd.grad if fd.grad is not None else torch.zeros(n, 1), fo.grad, rtol=1e-4, atol=1e-7)


def test_adam_matches_torch():
    from txt2vid_amd import functional as TF
    p, g1, g2 = rnd(1, 1000), rnd(2, 1000), rnd(3, 1000)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.5, 0.999))
    pd = p.to(dev())
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step, g in enumerate((g1, g2), 1):
        pr.grad = g.clone()
        opt.step()
        TF.adam_step(pd, g.to(dev()), m, v, 2e-4, 0.5, 0.999, 1e-8, step)
        close(pd, pr, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('momentum', [0.0, 0.5])
def test_sgd_matches_torch(momentum):
    """`txt2vid_amd.optim.SGD` (t2v_sgd_multi) == torch.optim.SGD(lr, momentum) over 3 steps, ragged sizes, a frozen parameter."""
    from txt2vid_amd.optim import SGD
    shapes = [(5,), (4099,), (3, 7, 11), (1,)]
    ps0 = [rnd(100 + i, *sh) for i, sh in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps0]
    got = [torch.nn.Parameter(p.clone().to(dev())) for p in ps0]
    o_ref, o_got = torch.optim.SGD(ref, lr=0.05, momentum=momentum), SGD(got, lr=0.05, momentum=momentum)
    for step in range(3):
        for i, (a, b) in enumerate(zip(ref, got)):
            if i == 3 and step == 1:
                a.grad = b.grad = None
                continue
            g = rnd(200 + 10 * step + i, *a.shape)
            a.grad, b.grad = g.clone(), g.to(dev())
        o_ref.step()
        o_got.step()
        for a, b in zip(ref, got):
            close(b, a, rtol=1e-6, atol=1e-6)
    if momentum:
        close(o_got.state[got[1]]['momentum_buffer'], o_ref.state[ref[1]]['momentum_buffer'], rtol=1e-6, atol=1e-6)


def test_pyramid_gather():
    from txt2vid_amd import functional as TF
    x = rnd(1, 5, 2, 8, 16, 16)
    for bt in (0, 1):
        y = TF.pyramid_gather(x.to(dev()), 3, 4, 16, 16, 2, 2, bt)
        close(y, x[::2, :, bt::2])
    y = TF.pyramid_gather(x.to(dev()), 5, 8, 4, 4, 1, 1, 0)
    close(y, F.interpolate(x, size=(8, 4, 4)))


@pytest.mark.parametrize('ta,tb', [(False, False), (True, False), (False, True), (True, True)])
def test_bmm_wide_tiles(ta, tb):
    """The 16 x 64-tile batched GEMM (N >= 64, N % 4 == 0) against torch.bmm on the host for every transposition, with ragged
    M / K (5, 33 rows; 4, 19, 70 deep) and N = 64, 100, 260; N = 66 keeps the 16 x 16 kernel."""
    from txt2vid_amd import functional as TF
    gen = torch.Generator()
    gen.manual_seed(11)
    for M, N, K in [(5, 64, 4), (33, 100, 19), (16, 260, 70), (33, 66, 19)]:
        A = torch.randn((3, K, M) if ta else (3, M, K), generator=gen)
        B = torch.randn((3, N, K) if tb else (3, K, N), generator=gen)
        want = torch.bmm(A.transpose(1, 2) if ta else A, B.transpose(1, 2) if tb else B)
        got = TF.bmm(A.to(dev()), B.to(dev()), ta, tb).cpu()
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)


def _pool_ref(x, w, b, stem):
    """pool(conv3(relu(x), w) + b) as the reference computes it: the stem's AvgPool3d((1,2,2), stride 2) (resnet3d.py:16) or
    DownSample's pooling by 2 on every extent > 1 (layers.py:202-215)."""
    y = F.conv3d(F.relu(x), w, b, padding=1)
    if stem:
        return F.avg_pool3d(y, (1, 2, 2), 2)
    k = tuple(2 if n > 1 else 1 for n in y.shape[2:])
    return F.avg_pool3d(y, k, k)


@pytest.mark.parametrize('stem,members,cin,cout', [
    (True, [(2, 4, 16, 16), (1, 2, 8, 8), (3, 1, 4, 4)], 64, 64),          # stem pooling: even frames, no time box; one member without a time axis
    (False, [(2, 4, 8, 8), (2, 2, 16, 16), (3, 1, 4, 4)], 64, 96),         # DownSample: (2,2,2) boxes, a (1,2,2) member, two channel tiles
    (False, [(2, 1, 32, 32), (1, 1, 2, 2)], 32, 64),                       # no member has a time axis: the 9-tap weight set
    (True, [(3, 6, 10, 6)], 96, 32),                                       # non-power-of-two extents, Cin = 96 (three channel blocks)
])
def test_pool_conv_group(stem, members, cin, cout):
    """`pool_conv_group` (box-sum + stride-2 GEMMs, functional_pool.py) against torch's avg_pool3d(conv3d(relu(x))) on the CPU:
    values, first-order gradients (data, weight, bias) and — like the gradient penalty — gradients THROUGH a recorded backward
    (d penalty / d x, d penalty / d w), i.e. all three GEMMs (forward, data gradient, weight gradient) in both of their roles."""
    from txt2vid_amd import functional as TF
    xs_h = [rnd(90 + i, n, cin, d, h, w) for i, (n, d, h, w) in enumerate(members)]
    w_h, b_h = rnd(7, cout, cin, 3, 3, 3) * 0.05, rnd(8, cout)
    xs = [t.to(dev()).requires_grad_(True) for t in xs_h]
    w, b = torch.nn.Parameter(w_h.to(dev())), torch.nn.Parameter(b_h.to(dev()))
    assert TF.pool_conv_ok(xs, w, stem)
    ys = TF.pool_conv_group(xs, w, b, relu_in=True, stem=stem)
    xr = [t.clone().requires_grad_(True) for t in xs_h]
    wr, br = w_h.clone().requires_grad_(True), b_h.clone().requires_grad_(True)
    yr = [_pool_ref(x, wr, br, stem) for x in xr]
    for y, r in zip(ys, yr):
        assert y.shape == r.shape
        close(y.detach(), r.detach())
    gys = [rnd(50 + i, *y.shape) for i, y in enumerate(yr)]

    def second_order(outs, leaves, to):
        f = sum((o * g.to(to)).sum() for o, g in zip(outs, gys))
        g1 = torch.autograd.grad(f, leaves, create_graph=True)
        pen = sum((g * g).sum() for g in g1[:len(outs)])                     # (penalty on the data gradients, like the GP)
        g2 = torch.autograd.grad(pen, leaves, allow_unused=True)
        return [g.detach() for g in g1], g2

    got1, got2 = second_order(ys, xs + [w, b], dev())
    ref1, ref2 = second_order(yr, xr + [wr, br], 'cpu')
    for a, r in zip(got1, ref1):
        close(a, r, rtol=2e-4, atol=2e-4)
    n = len(members)
    close(got2[n], ref2[n], rtol=2e-4, atol=2e-4)                           # d penalty / d w through the recorded data gradient
    for a, r in zip(got2[:n], ref2[:n]):                                    # d penalty / d x: zero almost everywhere (ReLU'' = 0), both sides
        assert (a is None or float(a.abs().max()) == 0.0) and (r is None or float(r.abs().max()) == 0.0)
    # first order WITHOUT a recorded graph (the plain backward of the D / G steps)
    ys = TF.pool_conv_group(xs, w, b, relu_in=True, stem=stem)
    plain = torch.autograd.grad(sum((o * g.to(dev())).sum() for o, g in zip(ys, gys)), xs + [w, b])
    for a, r in zip(plain, ref1):
        close(a, r, rtol=2e-4, atol=2e-4)


def test_pool_conv_group_at_benchmark_size():
    """The stem's pooled convolution over the 8 discriminator-step members at the benchmark size (M = 393 216 input voxels ->
    49 152 pooled rows) and down0's over its 8 members: forward, data gradient and weight + bias gradient (raw launches, k-split
    weight gradient with its reduce) against torch on the CPU."""
    from txt2vid_amd import functional as TF
    import conv_cases as cc
    for name, cin, cout, members, stem in cc.POOL_CASES[:2]:
        w = rnd(2, cout, cin, 3, 3, 3) * (1.0 / np.sqrt(cin * 27.0))
        b = rnd(3, cout) * 0.1
        xs = [rnd(10 + i, n, cin, d, h, wd) for i, (n, d, h, wd) in enumerate(members)]
        tmodes = [TF.pool_tmode(x.shape, stem) for x in xs]
        shapes = [tuple(x.shape) for x in xs]
        wd_, bd_ = w.to(dev()), b.to(dev())
        xd = [x.to(dev()) for x in xs]
        rts = TF.boxsum_raw(xd, tmodes, True)
        ys = TF.pool_fwd_raw(rts, shapes, tmodes, wd_, bd_)
        gys = [rnd(40 + i, *y.shape) for i, y in enumerate(ys)]
        gyd = [g.to(dev()) for g in gys]
        gxs = TF.unbox_raw(TF.pool_dgrad_raw(gyd, shapes, tmodes, wd_), shapes, tmodes, masks=xd)
        dbias = torch.empty(cout, device=dev())
        dw = TF.pool_wgrad_raw(rts, gyd, shapes, tmodes, tuple(w.shape), dbias=dbias)
        torch.cuda.synchronize()
        dw_ref = torch.zeros(w.shape, dtype=torch.float64)
        db_ref = torch.zeros(cout, dtype=torch.float64)
        for i, (x, gy) in enumerate(zip(xs, gys)):
            xr = x.clone().requires_grad_(True)
            wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            yr = _pool_ref(xr, wr, br, stem)
            close(ys[i], yr)
            (yr * gy).sum().backward()
            close(gxs[i], xr.grad)
            dw_ref += wr.grad.double()
            db_ref += br.grad.double()
        close(dw, dw_ref, rtol=2e-4, atol=2e-4)
        close(dbias, db_ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('n,cin,cout,h,w', [(3, 64, 32, 4, 4), (2, 128, 64, 8, 6), (5, 32, 32, 1, 1), (2, 96, 160, 16, 16)])
def test_up_conv(n, cin, cout, h, w):
    """`up_conv` (the pooled-convolution kernels in transposed roles, functional_pool.py): conv3x3(upsample2x(x)) + b of the
    generator's UpBlocks (layers.py:152-195) without the up-sampled tensor — values and first-order gradients (data, weight, bias)
    against torch's conv2d(interpolate(x, nearest x2)) on the CPU, with and without a gradient sink slot."""
    from txt2vid_amd import functional as TF
    x_h, w_h, b_h = rnd(1, n, cin, h, w), rnd(2, cout, cin, 3, 3) * 0.05, rnd(3, cout)
    xr, wr, br = x_h.clone().requires_grad_(True), w_h.clone().requires_grad_(True), b_h.clone().requires_grad_(True)
    yr = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, br, padding=1)
    gy = rnd(4, *yr.shape)
    (yr * gy).sum().backward()
    xd = x_h.to(dev()).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w_h.to(dev())), torch.nn.Parameter(b_h.to(dev()))
    assert TF.up_conv_ok(xd, wd)
    yd = TF.up_conv(xd, wd, bd)
    assert yd.shape == yr.shape
    close(yd, yr)
    (yd * gy.to(dev())).sum().backward()
    close(xd.grad, xr.grad)
    close(wd.grad, wr.grad, rtol=2e-4, atol=2e-4)
    close(bd.grad, br.grad, rtol=2e-4, atol=2e-4)
