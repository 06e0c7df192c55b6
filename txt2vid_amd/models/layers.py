"""Building blocks of the TGANv2 generator / discriminator, same class names, constructor
signatures, `forward` signatures and `state_dict` layout as txt2vid/models/layers.py — but every
tensor op is a hand-written gfx950 kernel reached through `txt2vid_amd.functional`.

Leaf layers subclass the torch layer they stand in for ONLY to inherit parameter creation (so that
seeding + `init()` reproduce the reference's initial weights and checkpoints interchange); their
`forward` never calls into ATen.
"""
import torch
import torch.nn as nn
from torch.nn import Parameter as P

from .. import functional as TF


def _check_same_conv(m):
    k = m.kernel_size
    if any(s != 1 for s in m.stride) or any(d != 1 for d in m.dilation) or m.groups != 1 or \
            any(p != kk // 2 for p, kk in zip(m.padding, k)) or any(kk not in (1, 3) for kk in k):
        raise NotImplementedError('HIP conv path covers stride-1 same-padded 1/3-wide kernels (all the hot path uses)')


class Conv3d(nn.Conv3d):
    """nn.Conv3d call sites: resnet3d.py:13-18, layers.py:231-238 (via which_conv)."""

    def forward(self, x):
        _check_same_conv(self)
        return TF.conv(x, self.weight, self.bias)


class Conv2d(nn.Conv2d):
    """nn.Conv2d call sites: layers.py:174-183,251, conv_lstm.py:19-26."""

    def forward(self, x):
        _check_same_conv(self)
        return TF.conv(x, self.weight, self.bias)


class Linear(nn.Linear):
    """nn.Linear call sites: resnet3d.py:33-35, tganv2*/gen.py fc."""

    def forward(self, x):
        return TF.linear(x, self.weight, self.bias)


class BatchNorm2d(nn.BatchNorm2d):
    """Training-mode batch statistics / eval-mode running statistics, optional fused ReLU."""

    def forward(self, x, relu=False, up=False, fork=False):
        """`fork`: returns (y, x') — x' is x for its second consumer (functional.BatchNormActFork)."""
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError('only the default BatchNorm2d configuration is on the hot path')
        return TF.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.momentum, self.eps, relu, counter=self.num_batches_tracked if self.training else None, up=up,
                                 fork=fork)


class ReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return TF.relu(x)


class Upsample(nn.Module):
    def __init__(self, scale_factor=2):
        super().__init__()
        assert scale_factor == 2
        self.scale_factor = scale_factor

    def forward(self, x):
        return TF.upsample2x(x)


class Tanh(nn.Module):
    def forward(self, x):
        return TF.tanh(x)


class AvgPool3d(nn.Module):
    """nn.AvgPool3d((1,2,2), 2) of resnet3d.py:16,18 — the int stride applies to T as well."""

    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        t3 = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)
        self.kernel_size = t3(kernel_size)
        self.stride = t3(stride if stride is not None else kernel_size)
        self.padding = t3(padding)

    def forward(self, x):
        return TF.avg_pool3d(x, self.kernel_size, self.stride, self.padding)


def _nonlocal(self, x, pooled_planes):
    """Shared body of Attention / Attention3d (layers.py:23-36, 52-68)."""
    b = x.size(0)
    x_res = x
    if x.dim() == 4 and x.is_cuda and x.requires_grad and torch.is_grad_enabled():
        # the block's input feeds three projections and the residual: the projections run as ONE Function whose data gradients
        # land in one buffer, and a grouped fork sums that with the residual's (else the autograd engine adds the four
        # contributions with three ATen launches)
        (xa,), (x_res,) = TF.fork_group([x])
        w5 = [m.weight.unsqueeze(2) for m in (self.theta, self.phi, self.g)]
        (theta,), (phi,), (g,) = TF.conv_multi_group([xa.unsqueeze(2)], [(w, None, False) for w in w5])
        theta, phi, g = theta.squeeze(2), TF.max_pool2x2(phi.squeeze(2)), TF.max_pool2x2(g.squeeze(2))
    else:
        theta = self.theta(x)
        phi = TF.max_pool2x2(self.phi(x))
        g = TF.max_pool2x2(self.g(x))
    theta = theta.reshape(b, self.ch // 8, -1)
    phi = phi.reshape(b, self.ch // 8, -1)
    g = g.reshape(b, self.ch // 2, -1)
    if getattr(self, 'fused_attend', False) and TF.nonlocal_attend_ok(self.ch // 8, self.ch // 2):
        # the generator's 2-D block: scores -> softmax -> weighted sum in one kernel, beta [b, N, N/4] never materialised
        o = TF.nonlocal_attend(theta, phi, g)
    else:
        beta = TF.softmax_lastdim(TF.bmm(theta, phi, True, False))          # [b, N, N/4]
        o = TF.bmm(g, beta, False, True)
    o = o.reshape((b, self.ch // 2) + tuple(x.shape[2:]))
    o = self.o(o)
    return TF.scale_add(self.gamma, o, x_res)


def nonlocal_levels(att, xs):
    """Attention3d over a list of pyramid levels: the four 1x1x1 convolutions are grouped launches over the levels, and
    so is every small op in between (max-pool, the two batched GEMMs, the softmax, gamma * o + x): 9 launches for all
    levels together (layers.py:52-68)."""
    if len(xs) > 8:
        raise ValueError('at most 8 tensors per grouped non-local block')
    xs, xs_res = TF.fork_group(xs)           # projections + residual: their gradients are summed in one launch for all levels
    thetas, phis, gs = TF.conv_multi_group(xs, [(att.theta.weight, None, False), (att.phi.weight, None, False),
                                                (att.g.weight, None, False)])
    phis = TF.max_pool2x2_group(phis)
    gs = TF.max_pool2x2_group(gs)
    bs = [x.size(0) for x in xs]
    thetas = [t.reshape(b, att.ch // 8, -1) for t, b in zip(thetas, bs)]
    phis = [t.reshape(b, att.ch // 8, -1) for t, b in zip(phis, bs)]
    gs = [t.reshape(b, att.ch // 2, -1) for t, b in zip(gs, bs)]
    betas = TF.softmax_lastdim_group(TF.bmm_group(thetas, phis, True, False))
    os_ = TF.bmm_group(gs, betas, False, True)
    os_ = [o.reshape((b, att.ch // 2) + tuple(x.shape[2:])) for o, b, x in zip(os_, bs, xs)]
    os_ = TF.conv_group(os_, att.o.weight, None)
    return TF.scale_add_group(att.gamma, os_, xs_res)


class Attention(nn.Module):
    """2-D non-local block (SA-GAN style) — layers.py:10-36."""

    def __init__(self, ch, which_conv=None, name='attention'):
        super().__init__()
        which_conv = which_conv or Conv2d
        self.ch = ch
        self.which_conv = which_conv
        self.theta = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = which_conv(ch, ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = which_conv(ch // 2, ch, kernel_size=1, padding=0, bias=False)
        self.gamma = P(torch.tensor(0.), requires_grad=True)
        self.fused_attend = True          # first-order use only (generator): the fused kernel has no second-order adjoint

    def forward(self, x, y=None):
        return _nonlocal(self, x, 2)


class Attention3d(nn.Module):
    """3-D non-local block, max-pool [1,2,2] on phi/g — layers.py:39-68."""

    def __init__(self, ch, which_conv=None, name='attention'):
        super().__init__()
        which_conv = which_conv or Conv3d
        self.ch = ch
        self.which_conv = which_conv
        self.theta = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = which_conv(ch, ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = which_conv(ch // 2, ch, kernel_size=1, padding=0, bias=False)
        self.gamma = P(torch.tensor(0.), requires_grad=True)

    def forward(self, x, y=None):
        return _nonlocal(self, x, 3)


class Identity(nn.Module):
    def forward(self, x):
        return x


class ResidualBlock(nn.Module):
    """identity_map(x) + inner_module(x); tags the inner path `is_residual` so that `init` applies the
    sqrt(2) gain — layers.py:77-96."""

    def __init__(self, inner_module=None, identity_map=None):
        super().__init__()
        self.inner_module = inner_module
        self.identity_map = identity_map if identity_map is not None else Identity()

        def tag(m):
            m.is_residual = True
        self.inner_module.apply(tag)

    def forward(self, x):
        return TF.add(self.identity_map(x), self.inner_module(x))


class Subsample(nn.Module):
    """`x[::sn, :, bt::st]`, bt ~ randint(st) on the CPU generator — layers.py:98-111."""

    def __init__(self, sn=2, st=2):
        super().__init__()
        self.sn, self.st = sn, st

    def forward(self, x, bt=None):
        if bt is None:
            bt = torch.randint(self.st, (1,))
        bt_i = int(bt)
        if x.is_cuda and x.dim() == 5 and self.sn == 2 and self.st == 2:
            return TF.PyramidGather.apply(x, bt_i), bt
        return x[::self.sn, :, bt_i::self.st], bt


class UpBlock(nn.Module):
    """BN-ReLU-Up-conv3x3-BN-ReLU-conv3x3 (+) Up-[conv1x1] — layers.py:152-195."""

    def __init__(self, in_channels=128, out_channels=None, which_bn=None, which_conv=None, upsample_instead=True,
                 which_unpool=None, wide=False, with_non_local=False):
        super().__init__()
        which_bn = which_bn or BatchNorm2d
        which_conv = which_conv or Conv2d
        self.in_channels = in_channels
        self.out_channels = out_channels = in_channels if out_channels is None else out_channels
        mid = in_channels if wide else out_channels
        assert upsample_instead
        main = nn.Sequential(which_bn(in_channels), ReLU(), Upsample(2), which_conv(in_channels, mid, 3, 1, padding=1),
                             which_bn(mid), ReLU(), which_conv(mid, out_channels, 3, 1, padding=1))
        ident = Upsample(2)
        if in_channels != out_channels:
            ident = nn.Sequential(ident, which_conv(in_channels, out_channels, 1))
        self.main = ResidualBlock(inner_module=main, identity_map=ident)
        self.with_non_local = with_non_local
        if with_non_local:
            self.attn = Attention(out_channels)

    def forward(self, x):
        m = self.main.inner_module
        # (the block input feeds the BatchNorm and the skip path: the BatchNorm adjoint sums the skip path's gradient in its own pass)
        if isinstance(m[0], BatchNorm2d) and isinstance(m[2], Upsample) and isinstance(m[3], Conv2d) and TF.up_conv_ok(x, m[3].weight):
            # Up -> conv3x3 without the up-sampled tensor: 9 taps per INPUT pixel (a quarter of the MACs), functional_pool.py
            _check_same_conv(m[3])
            h, x = m[0](x, relu=True, fork=True)
            h = TF.up_conv(h, m[3].weight, m[3].bias)
        elif isinstance(m[0], BatchNorm2d) and isinstance(m[2], Upsample):
            h, x = m[0](x, relu=True, up=True, fork=True)                              # BN+ReLU+Up fused
            h = m[3](h)
        else:
            h = m[0](x, relu=True) if isinstance(m[0], BatchNorm2d) else m[1](m[0](x))
            h = m[3](m[2](h))
        h = m[4](h, relu=True) if isinstance(m[4], BatchNorm2d) else m[5](m[4](h))
        h = m[6](h)
        idm = self.main.identity_map
        if isinstance(idm, Upsample):
            x = TF.upsample_add(x, h)                                   # Up(x) + h in one launch
        elif isinstance(idm, nn.Sequential) and len(idm) == 2 and isinstance(idm[0], Upsample) and isinstance(idm[1], Conv2d) \
                and tuple(idm[1].kernel_size) == (1, 1):
            # conv1x1(Up(x)) == Up(conv1x1(x)) value for value (a 1x1 convolution is per-pixel): run it on the small map
            x = TF.upsample_add(idm[1](x), h)
        else:
            x = TF.add(idm(x), h)
        if self.with_non_local:
            x = self.attn(x)
        return x


import os as _os
SKIP_POOLS_FIRST = _os.environ.get('T2V_NO_SKIP_POOLS_FIRST') is None        # developer A/B switch (see down_block_levels)
SKIP_POOLS_FIRST_MIN_VOXELS = 16384       # below this a block's launches are latency-bound: the two extra small launches cost more


def downsample_cfg(x):
    """kernel / stride / padding of `DownSample` for this tensor — layers.py:202-215."""
    k, s, p = [1, 1, 1], [1, 1, 1], [0, 0, 0]
    for i in range(3):
        n = x.size(i + 2)
        if n == 1:
            continue
        k[i] = s[i] = 2
        if n % 2:
            p[i] = 1
    return k, s, p


def downsample_sum(a, b):
    """DownSample(a) + DownSample(b) == DownSample(a + b) (average pooling is linear): one fused kernel."""
    k, s, p = downsample_cfg(a)
    if k == [1, 1, 1]:
        return TF.add(a, b)
    return TF.add_avg_pool3d(a, b, k, s, p)


class DownSample(nn.Module):
    """Average-pool by 2 every spatial/temporal dim of extent > 1 (pad 1 if odd) — layers.py:197-217."""

    def forward(self, x):
        k, s, p = downsample_cfg(x)
        if k == [1, 1, 1]:
            return x
        return TF.avg_pool3d(x, k, s, p)


class DownBlock(nn.Module):
    """ReLU-conv-ReLU-conv-DownSample (+) conv1-DownSample — layers.py:219-243."""

    def __init__(self, in_channels=3, out_channels=None, which_conv=None, wide=True):
        super().__init__()
        which_conv = which_conv or Conv3d
        out_channels = in_channels if out_channels is None else out_channels
        mid = out_channels if wide else in_channels
        main = nn.Sequential(ReLU(), which_conv(in_channels, mid, kernel_size=3, padding=1), ReLU(),
                             which_conv(mid, out_channels, kernel_size=3, padding=1), DownSample())
        ident = nn.Sequential(which_conv(in_channels, out_channels, 1), DownSample())
        self.main = ResidualBlock(inner_module=main, identity_map=ident)

    def forward(self, x):
        m = self.main.inner_module
        if isinstance(m[1], Conv3d) and isinstance(m[3], Conv3d):
            idm = self.main.identity_map
            if x.is_cuda and isinstance(idm[0], Conv3d):
                return down_block_levels(self, [x])[0]             # a group of one: the same launches as the multi-level pass
            h = TF.relu_conv(x, m[1].weight, m[1].bias)            # ReLU fused into the conv gather
            if isinstance(idm[1], DownSample) and isinstance(m[4], DownSample) and TF.pool_conv_ok([h], m[3].weight, False):
                # conv2 -> DownSample as ONE pooled convolution; the skip path's pooling adds it in its launch (as `down_block_levels`)
                z = TF.pool_conv_group([h], m[3].weight, m[3].bias, relu_in=True, stem=False)[0]
                s_ = idm[0](x)
                return TF.avg_pool3d_group([s_], [downsample_cfg(s_)], adds=[z])[0]
            h = TF.relu_conv(h, m[3].weight, m[3].bias)
            if isinstance(idm[1], DownSample) and isinstance(m[4], DownSample):
                return downsample_sum(idm[0](x), h)
            return TF.add(idm(x), m[4](h))
        return self.main(x)


def down_block_levels(block, xs):
    """DownBlock over a list of pyramid levels, layer by layer: each convolution is one grouped launch."""
    m = block.main.inner_module
    idm = block.main.identity_map
    cfgs = [downsample_cfg(x) for x in xs]
    if SKIP_POOLS_FIRST and isinstance(idm[1], DownSample) and isinstance(m[4], DownSample) and isinstance(idm[0], Conv3d) and \
            tuple(idm[0].weight.shape[2:]) == (1, 1, 1) and not any(any(p) for _, _, p in cfgs) and \
            sum(x.numel() // x.shape[1] for x in xs) >= SKIP_POOLS_FIRST_MIN_VOXELS and TF.pool_conv_ok(xs, m[3].weight, False):
        # Round 4: the skip path `conv1x1x1 -> DownSample` (layers.py:233-238) runs as DownSample -> conv1x1x1. A 1x1x1 convolution
        # acts per voxel and un-padded average pooling is linear, so the two commute value for value up to summation order (bias
        # included: the mean of a constant); the convolution then runs on an EIGHTH of the voxels, its two adjoints likewise, and
        # the pooling moves C_in channels instead of C_out. The block's input feeds two consumers: a grouped fork sums their
        # gradients in one launch (as in the stem). Only where the block's tensors are big enough for the saved work to show.
        xa, xp = TF.fork_pool_group(xs, cfgs)         # (the pooling adjoint adds the main path's gradient in its own launch)
        hs = TF.conv_group(xa, m[1].weight, m[1].bias, relu_in=True)
        zs = TF.pool_conv_group(hs, m[3].weight, m[3].bias, relu_in=True, stem=False)
        ss = TF.conv_group(xp, idm[0].weight, idm[0].bias)
        return TF.add_group(zs, ss)
    # main conv1 and the skip conv read the same tensors: one Function, so their data gradients land in one buffer
    hs, ss = TF.conv_multi_group(xs, [(m[1].weight, m[1].bias, True), (idm[0].weight, idm[0].bias, False)])
    if isinstance(idm[1], DownSample) and isinstance(m[4], DownSample) and TF.pool_conv_ok(hs, m[3].weight, False):
        # the second convolution feeds ONLY DownSample: pool(conv3(relu(h))) as the stride-2 convolution of the box-summed
        # activation (an eighth of the MACs where all three extents are pooled); the skip path's pooling adds it in its launch
        zs = TF.pool_conv_group(hs, m[3].weight, m[3].bias, relu_in=True, stem=False)
        return TF.avg_pool3d_group(ss, [downsample_cfg(s_) for s_ in ss], adds=zs)
    hs = TF.conv_group(hs, m[3].weight, m[3].bias, relu_in=True)
    if isinstance(idm[1], DownSample) and isinstance(m[4], DownSample):
        return TF.avg_pool3d_group(ss, [downsample_cfg(s_) for s_ in ss], x2s=hs)     # one launch for all levels
    return [TF.add(idm[1](s_), m[4](h)) for s_, h in zip(ss, hs)]


class RenderBlock(nn.Module):
    """BN-ReLU-conv3x3-tanh — layers.py:245-259."""

    def __init__(self, in_channels=128, out_channels=3, which_bn=None, which_conv=None):
        super().__init__()
        which_bn = which_bn or BatchNorm2d
        which_conv = which_conv or Conv2d
        self.bn = which_bn(in_channels)
        self.activation = ReLU()
        self.conv = which_conv(in_channels, out_channels, kernel_size=3, padding=1)
        self.final = Tanh()

    def forward(self, x, fork=False):
        """`fork`: returns (image, x') — x' is the map for the next level (its gradient joins the BatchNorm adjoint's pass)."""
        if fork and isinstance(self.bn, BatchNorm2d):
            h, x = self.bn(x, relu=True, fork=True)
            return self.final(self.conv(h)), x
        h = self.bn(x, relu=True) if isinstance(self.bn, BatchNorm2d) else self.activation(self.bn(x))
        r = self.final(self.conv(h))
        return (r, x) if fork else r
