"""CPU oracle for the TGANv2 training hot path (TEST INFRASTRUCTURE — never shipped, never imported
by the product package `txt2vid_amd/`; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file).

It is a *restatement*, in plain fp32 PyTorch CPU ops, of the reference algorithm for the path
`BASELINE.json:north_star` names.  It is written functionally: every network is a pure function of a
flat ``dict[str, Tensor]`` that uses the reference's own ``state_dict`` key names, so the same weights
can be poured into (a) the imported reference (in the build container, by
`tests/golden/make_golden.py`), (b) this oracle and (c) the HIP-backed modules in `txt2vid_amd/`.

Parity pin: `tests/golden/*.npz` are produced by `tests/golden/make_golden.py`, which imports the
real reference from /root/reference (build container only) and records its outputs; the CPU test
`tests/test_oracle_golden.py` checks every function here against those vectors.

Reference citations (relative to /root/reference/) are given per function.
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# weights "by recipe" (SURVEY.md §8c): deterministic per state_dict key, so 345 MB of generator
# weights never have to be stored in a fixture.
# --------------------------------------------------------------------------------------------

def recipe_tensor(key, shape, base_seed=0, attn_gamma=0.5):
    """Deterministic tensor for state_dict entry `key`.

    weights (dim >= 2): N(0,1) * sqrt(2 / (fan_in + fan_out))   (xavier-normal scale)
    1-d `.weight` (BatchNorm gamma): 1 + 0.1 N(0,1);  `.bias`: 0.05 N(0,1)
    `running_mean`: 0.05 N(0,1); `running_var`: 1 + 0.1 U(0,1); `num_batches_tracked`: 0
    `gamma` (non-local gain, 0-d): `attn_gamma` (non-zero so the non-local path is exercised)
    """
    shape = tuple(shape)
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ base_seed) & 0x7FFFFFFF)
    if key.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith('gamma') and len(shape) == 0:
        return torch.tensor(float(attn_gamma))
    if key.endswith('running_var'):
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if key.endswith('running_mean'):
        return 0.05 * torch.randn(shape, generator=g)
    if len(shape) >= 2:
        rf = 1
        for s in shape[2:]:
            rf *= s
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / (fan_in + fan_out))
    if key.endswith('weight'):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)


def recipe_state(shapes, base_seed=0, attn_gamma=0.5):
    """shapes: dict key -> shape. Returns dict key -> tensor (fp32 / long)."""
    return {k: recipe_tensor(k, s, base_seed, attn_gamma) for k, s in shapes.items()}


# --------------------------------------------------------------------------------------------
# shape tables (what `state_dict()` of the reference modules holds; [probed] in SURVEY §8b)
# --------------------------------------------------------------------------------------------

def resnet3d_shapes(prefix='', num_channels=1, mid_ch=64, cond_dim=0, num_down_blocks=4, with_attn=True):
    """Keys of `Resnet3D.state_dict()` — txt2vid/models/resnet3d.py:8-35."""
    s = {}
    p = prefix
    s[p + 'res_block.inner_module.0.weight'] = (mid_ch, num_channels, 3, 3, 3)
    s[p + 'res_block.inner_module.0.bias'] = (mid_ch,)
    s[p + 'res_block.inner_module.2.weight'] = (mid_ch, mid_ch, 3, 3, 3)
    s[p + 'res_block.inner_module.2.bias'] = (mid_ch,)
    s[p + 'res_block.identity_map.1.weight'] = (mid_ch, num_channels, 1, 1, 1)
    s[p + 'res_block.identity_map.1.bias'] = (mid_ch,)
    in_ch, out_ch, idx = mid_ch, 128, 0
    for i in range(num_down_blocks):
        q = p + 'down.%d.' % idx
        s[q + 'main.inner_module.1.weight'] = (in_ch, in_ch, 3, 3, 3)     # wide=False -> mid=in
        s[q + 'main.inner_module.1.bias'] = (in_ch,)
        s[q + 'main.inner_module.3.weight'] = (out_ch, in_ch, 3, 3, 3)
        s[q + 'main.inner_module.3.bias'] = (out_ch,)
        s[q + 'main.identity_map.0.weight'] = (out_ch, in_ch, 1, 1, 1)
        s[q + 'main.identity_map.0.bias'] = (out_ch,)
        idx += 1
        if i == 0 and with_attn:
            q = p + 'down.%d.' % idx
            s[q + 'gamma'] = ()
            s[q + 'theta.weight'] = (out_ch // 8, out_ch, 1, 1, 1)
            s[q + 'phi.weight'] = (out_ch // 8, out_ch, 1, 1, 1)
            s[q + 'g.weight'] = (out_ch // 2, out_ch, 1, 1, 1)
            s[q + 'o.weight'] = (out_ch, out_ch // 2, 1, 1, 1)
            idx += 1
        in_ch, out_ch = out_ch, out_ch * 2
    s[p + 'fc_uncond.weight'] = (1, in_ch)
    s[p + 'fc_uncond.bias'] = (1,)
    if cond_dim > 0:
        s[p + 'fc.weight'] = (1, in_ch + cond_dim)
        s[p + 'fc.bias'] = (1,)
    return s


def _bn_shapes(s, q, c):
    s[q + 'weight'] = (c,)
    s[q + 'bias'] = (c,)
    s[q + 'running_mean'] = (c,)
    s[q + 'running_var'] = (c,)
    s[q + 'num_batches_tracked'] = ()


def upblock_shapes(prefix, cin, cout, with_non_local=False):
    """Keys of `UpBlock.state_dict()` — txt2vid/models/layers.py:155-189 (wide=False: mid=out)."""
    s = {}
    q = prefix + 'main.inner_module.'
    _bn_shapes(s, q + '0.', cin)
    s[q + '3.weight'] = (cout, cin, 3, 3)
    s[q + '3.bias'] = (cout,)
    _bn_shapes(s, q + '4.', cout)
    s[q + '6.weight'] = (cout, cout, 3, 3)
    s[q + '6.bias'] = (cout,)
    if cin != cout:
        s[prefix + 'main.identity_map.1.weight'] = (cout, cin, 1, 1)
        s[prefix + 'main.identity_map.1.bias'] = (cout,)
    if with_non_local:
        a = prefix + 'attn.'
        s[a + 'gamma'] = ()
        s[a + 'theta.weight'] = (cout // 8, cout, 1, 1)
        s[a + 'phi.weight'] = (cout // 8, cout, 1, 1)
        s[a + 'g.weight'] = (cout // 2, cout, 1, 1)
        s[a + 'o.weight'] = (cout, cout // 2, 1, 1)
    return s


def gen_shapes(latent_size=256, width=64, height=64, num_channels=1, additional_blocks=(64, 32, 32),
               fm_channels=1024, cond_dim=0, cond_variant=False):
    """Keys of `MultiScaleGen.state_dict()` — txt2vid/models/tganv2/gen.py:24-60 and
    txt2vid/models/tganv2_cond/gen.py:24-62 (`cond_variant`: non-local block after block 2)."""
    s = {}
    fw, fh = max(1, width // 64), max(1, height // 64)
    s['fc.weight'] = (fm_channels * fw * fh, latent_size + cond_dim)
    s['fc.bias'] = (fm_channels * fw * fh,)
    for gate in 'ifco':
        s['clstm.cell0.Wx%s.weight' % gate] = (fm_channels, fm_channels, 3, 3)
        s['clstm.cell0.Wx%s.bias' % gate] = (fm_channels,)
        s['clstm.cell0.Wh%s.weight' % gate] = (fm_channels, fm_channels, 3, 3)
    s.update(upblock_shapes('abstract_blocks.0.up0.', fm_channels, 512))
    s.update(upblock_shapes('abstract_blocks.0.up1.', 512, 256))
    s.update(upblock_shapes('abstract_blocks.0.up2.', 256, 128))
    _bn_shapes(s, 'render_blocks.0.bn.', 128)
    s['render_blocks.0.conv.weight'] = (num_channels, 128, 3, 3)
    s['render_blocks.0.conv.bias'] = (num_channels,)
    prev = 128
    for i, ch in enumerate(additional_blocks):
        nl = cond_variant and (i == len(additional_blocks) - 2)
        s.update(upblock_shapes('abstract_blocks.%d.' % (i + 1), prev, ch, with_non_local=nl))
        _bn_shapes(s, 'render_blocks.%d.bn.' % (i + 1), ch)
        s['render_blocks.%d.conv.weight' % (i + 1)] = (num_channels, ch, 3, 3)
        s['render_blocks.%d.conv.bias' % (i + 1)] = (num_channels,)
        prev = ch
    return s


# --------------------------------------------------------------------------------------------
# discriminator — txt2vid/models/resnet3d.py, txt2vid/models/layers.py
# --------------------------------------------------------------------------------------------

def downsample(x):
    """`DownSample.forward` — txt2vid/models/layers.py:202-217: average-pool by 2 along every one of
    (T,H,W) whose extent is > 1 (pad 1 when odd; padded zeros count in the divisor)."""
    k, st, pd = [1, 1, 1], [1, 1, 1], [0, 0, 0]
    for i in range(3):
        n = x.size(i + 2)
        if n == 1:
            continue
        k[i], st[i] = 2, 2
        if n % 2:
            pd[i] = 1
    return F.avg_pool3d(x, kernel_size=k, stride=st, padding=pd)


def nonlocal3d(P, q, x):
    """`Attention3d.forward` — txt2vid/models/layers.py:52-68."""
    b, ch = x.size(0), x.size(1)
    theta = F.conv3d(x, P[q + 'theta.weight'])
    phi = F.max_pool3d(F.conv3d(x, P[q + 'phi.weight']), [1, 2, 2])
    g = F.max_pool3d(F.conv3d(x, P[q + 'g.weight']), [1, 2, 2])
    theta = theta.view(b, ch // 8, -1)
    phi = phi.view(b, ch // 8, -1)
    g = g.view(b, ch // 2, -1)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = torch.bmm(g, beta.transpose(1, 2)).view(b, -1, x.shape[2], x.shape[3], x.shape[4])
    o = F.conv3d(o, P[q + 'o.weight'])
    return P[q + 'gamma'] * o + x


def nonlocal2d(P, q, x):
    """`Attention.forward` — txt2vid/models/layers.py:23-36."""
    ch = x.size(1)
    hw = x.shape[2] * x.shape[3]
    theta = F.conv2d(x, P[q + 'theta.weight'])
    phi = F.max_pool2d(F.conv2d(x, P[q + 'phi.weight']), [2, 2])
    g = F.max_pool2d(F.conv2d(x, P[q + 'g.weight']), [2, 2])
    theta = theta.view(-1, ch // 8, hw)
    phi = phi.view(-1, ch // 8, hw // 4)
    g = g.view(-1, ch // 2, hw // 4)
    beta = F.softmax(torch.bmm(theta.transpose(1, 2), phi), -1)
    o = torch.bmm(g, beta.transpose(1, 2)).view(-1, ch // 2, x.shape[2], x.shape[3])
    o = F.conv2d(o, P[q + 'o.weight'])
    return P[q + 'gamma'] * o + x


def down_block(P, q, x):
    """`DownBlock.forward` — txt2vid/models/layers.py:229-243."""
    m = 'main.inner_module.'
    h = F.conv3d(F.relu(x), P[q + m + '1.weight'], P[q + m + '1.bias'], padding=1)
    h = F.conv3d(F.relu(h), P[q + m + '3.weight'], P[q + m + '3.bias'], padding=1)
    h = downsample(h)
    s = F.conv3d(x, P[q + 'main.identity_map.0.weight'], P[q + 'main.identity_map.0.bias'])
    return downsample(s) + h


def resnet3d(P, x, cond=None, prefix='', num_down_blocks=4, with_attn=True):
    """`Resnet3D.forward` — txt2vid/models/resnet3d.py:38-57. Returns (uncond, cond|None, feat)."""
    p = prefix
    r = 'res_block.inner_module.'
    h = F.conv3d(x, P[p + r + '0.weight'], P[p + r + '0.bias'], padding=1)
    h = F.conv3d(F.relu(h), P[p + r + '2.weight'], P[p + r + '2.bias'], padding=1)
    h = F.avg_pool3d(h, (1, 2, 2), 2)                       # stride 2 on T too (resnet3d.py:16)
    s = F.avg_pool3d(x, (1, 2, 2), 2)
    s = F.conv3d(s, P[p + 'res_block.identity_map.1.weight'], P[p + 'res_block.identity_map.1.bias'])
    h = s + h
    idx = 0
    for i in range(num_down_blocks):
        h = down_block(P, p + 'down.%d.' % idx, h)
        idx += 1
        if i == 0 and with_attn:
            h = nonlocal3d(P, p + 'down.%d.' % idx, h)
            idx += 1
    feat = torch.sum(h, [2, 3, 4])
    uncond = F.linear(feat, P[p + 'fc_uncond.weight'], P[p + 'fc_uncond.bias'])
    c = None
    if cond is not None:
        c = F.linear(torch.cat((feat, cond), dim=1), P[p + 'fc.weight'], P[p + 'fc.bias'])
    return uncond, c, feat


def multiscale_discrim(P, xs, conds=None, prefix='single_discrim.'):
    """`MultiScaleDiscrim.forward` with `single_discrim=True` (weights shared by the 4 levels) —
    txt2vid/models/tganv2/discrim.py:23-31, txt2vid/models/tganv2_cond/discrim.py:28-48
    (cond variant uses prefix 'single_discrim.module.')."""
    out = []
    for i, x in enumerate(xs):
        c = conds[i] if conds is not None else None
        out.append(resnet3d(P, x, c, prefix=prefix))
    return out


# --------------------------------------------------------------------------------------------
# generator — txt2vid/models/conv_lstm.py, txt2vid/models/layers.py, txt2vid/models/tganv2*/gen.py
# --------------------------------------------------------------------------------------------

def batchnorm_train(P, q, x, training=True):
    """nn.BatchNorm2d in train mode (batch statistics, running stats updated with momentum 0.1,
    eps 1e-5) — the `which_bn` of txt2vid/models/layers.py:171,175,249."""
    y = F.batch_norm(x, P[q + 'running_mean'], P[q + 'running_var'], P[q + 'weight'], P[q + 'bias'],
                     training, 0.1, 1e-5)
    if training:
        P[q + 'num_batches_tracked'] += 1
    return y


def up_block(P, q, x, training=True):
    """`UpBlock.forward` — txt2vid/models/layers.py:170-195."""
    m = q + 'main.inner_module.'
    h = F.relu(batchnorm_train(P, m + '0.', x, training))
    h = F.interpolate(h, scale_factor=2)                                   # nn.Upsample (nearest)
    h = F.conv2d(h, P[m + '3.weight'], P[m + '3.bias'], padding=1)
    h = F.relu(batchnorm_train(P, m + '4.', h, training))
    h = F.conv2d(h, P[m + '6.weight'], P[m + '6.bias'], padding=1)
    s = F.interpolate(x, scale_factor=2)
    if (q + 'main.identity_map.1.weight') in P:
        s = F.conv2d(s, P[q + 'main.identity_map.1.weight'], P[q + 'main.identity_map.1.bias'])
    y = s + h
    if (q + 'attn.gamma') in P:
        y = nonlocal2d(P, q + 'attn.', y)
    return y


def render_block(P, q, x, training=True):
    """`RenderBlock.forward` — txt2vid/models/layers.py:254-259."""
    h = F.relu(batchnorm_train(P, q + 'bn.', x, training))
    return torch.tanh(F.conv2d(h, P[q + 'conv.weight'], P[q + 'conv.bias'], padding=1))


def conv_lstm(P, q, x, steps=16):
    """`ConvLSTM.forward` with one cell — txt2vid/models/conv_lstm.py:75-97 and cell :32-38.
    The input is `x` at step 0 and zeros afterwards (:79); the peephole terms Wci/Wcf/Wco are
    constant zeros (:47-49) so they are dropped."""
    def cv(name, t, bias=True):
        return F.conv2d(t, P[q + name + '.weight'], P[q + name + '.bias'] if bias else None, padding=1)
    h = torch.zeros_like(x)
    c = torch.zeros_like(x)
    outs = []
    for step in range(steps):
        xin = x if step == 0 else torch.zeros_like(x)
        ci = torch.sigmoid(cv('Wxi', xin) + cv('Whi', h, False))
        cf = torch.sigmoid(cv('Wxf', xin) + cv('Whf', h, False))
        cc = cf * c + ci * torch.tanh(cv('Wxc', xin) + cv('Whc', h, False))
        co = torch.sigmoid(cv('Wxo', xin) + cv('Who', h, False))
        h = co * torch.tanh(cc)
        c = cc
        outs.append(h)
    return outs


def subsample(x, bt=None):
    """`Subsample.forward` — txt2vid/models/layers.py:106-111: `x[::2, :, bt::2]` on [B,C,T,H,W];
    `bt ~ randint(2)` drawn from the *global CPU* torch generator when not given."""
    if bt is None:
        bt = int(torch.randint(2, (1,)))
    return x[::2, :, bt::2], bt


def multiscale_gen(P, z, cond=None, training=True, num_frames=16, n_blocks=4, bts=None,
                   fm_channels=1024, output_blocks=None):
    """`MultiScaleGen.forward` — txt2vid/models/tganv2_cond/gen.py:64-124 (uncond twin
    txt2vid/models/tganv2/gen.py:62-119). Returns the list of rendered videos [b,C,T,H,W].
    `bts`: optional list of the 3 subsample phases; else drawn like the reference."""
    x = z if cond is None else torch.cat((z, cond), dim=1)
    x = F.linear(x, P['fc.weight'], P['fc.bias'])
    hw = x.size(1) // fm_channels
    fh = fw = int(round(math.sqrt(hw)))
    x = x.view(x.size(0), fm_channels, fh, fw)
    frames = conv_lstm(P, 'clstm.cell0.', x, num_frames)
    x = torch.stack(frames).permute(1, 0, 2, 3, 4)              # [B,T,C,h,w]
    T = num_frames

    def merge(a):
        return a.contiguous().view(-1, a.size(2), a.size(3), a.size(4))

    def split(a, t):
        return a.contiguous().view(-1, t, a.size(1), a.size(2), a.size(3))

    x = merge(x)
    rendered = []
    used_bts = []
    for i in range(n_blocks):
        if i != 0 and training:
            v = split(x, T).permute(0, 2, 1, 3, 4)
            v, bt = subsample(v, None if bts is None else bts[i - 1])
            used_bts.append(bt)
            x = merge(v.permute(0, 2, 1, 3, 4))
            T //= 2
        if i == 0:
            for u in ('up0.', 'up1.', 'up2.'):
                x = up_block(P, 'abstract_blocks.0.' + u, x, training)
        else:
            x = up_block(P, 'abstract_blocks.%d.' % i, x, training)
        if i == n_blocks - 1 or training or (output_blocks is not None and i in output_blocks):
            r = render_block(P, 'render_blocks.%d.' % i, x, training)
            rendered.append(split(r, T).permute(0, 2, 1, 3, 4))
    return rendered


# --------------------------------------------------------------------------------------------
# losses — txt2vid/gan/losses.py
# --------------------------------------------------------------------------------------------

def rsgan_discrim_loss(fake, real):
    """`RSGANLoss.discrim_loss` — txt2vid/gan/losses.py:79-81: BCEWithLogits(real-fake, 1)."""
    return F.binary_cross_entropy_with_logits(real - fake, torch.ones_like(fake))


def rsgan_gen_loss(fake, real):
    """`RSGANLoss.gen_loss` — txt2vid/gan/losses.py:83-85: BCEWithLogits(fake-real, 1)."""
    return F.binary_cross_entropy_with_logits(fake - real, torch.ones_like(fake))


def zoo_loss(kind, side, fake, real, margin=2.0):
    """The rest of txt2vid/gan/losses.py on D's logits, numpy-style closed forms (side 0 = discrim_loss, 1 = gen_loss):
    vanilla (:19-46; `LabelledGanLoss` stores its labels crossed at :27-28, so D sees fake->1, real->0 and G fake->0),
    hinge (:48-52, `HingeEmbeddingLoss(margin)`: label 1 -> mean x, label -1 -> mean relu(margin - x); fake->1, real->-1),
    wasserstein (:55-68), rasgan (:87-110 with the labels it means, 0/1 — as written it raises AttributeError),
    ralsgan (:113-133)."""
    sp = F.softplus
    if kind == 'vanilla':
        return sp(-fake).mean() + sp(real).mean() if side == 0 else sp(fake).mean()
    if kind == 'hinge':
        return fake.mean() + F.relu(margin - real).mean() if side == 0 else F.relu(margin - fake).mean()
    if kind == 'wasserstein':
        return -(real.mean() - fake.mean()) if side == 0 else -fake.mean()
    u, v = real - fake.mean(), fake - real.mean()
    if kind == 'rasgan':
        return (sp(-u).mean() + sp(v).mean()) / 2 if side == 0 else (sp(u).mean() + sp(-v).mean()) / 2
    if kind == 'ralsgan':
        return (((u - 1) ** 2).mean() + ((v + 1) ** 2).mean()) / 2 if side == 0 else (((u + 1) ** 2).mean() + ((v - 1) ** 2).mean()) / 2
    raise KeyError(kind)


def gp_level(P, prefix, real_x, fake_x, real_c=None, fake_c=None, alpha=None):
    """`_gradient_penalty(..., zero_center=True, combine=torch.sum)` for one pyramid level —
    txt2vid/gan/losses.py:135-186 as called from :203. `alpha` [b] ~ U[0,1) from the global CPU
    generator when not given (:140-145). Returns sum_b ||d(sum u + sum c)/d xhat_b||^2."""
    b = real_x.size(0)
    if alpha is None:
        alpha = torch.rand(b, 1, 1, 1, 1)
    alpha = alpha.view(b, 1, 1, 1, 1).to(real_x.dtype)
    xh = (alpha * real_x + (1 - alpha) * fake_x).detach().requires_grad_(True)
    ch = None
    if real_c is not None and fake_c is not None:
        a2 = alpha.view(b, 1)
        ch = a2 * real_c + (1 - a2) * fake_c
    u, c, _ = resnet3d(P, xh, ch, prefix=prefix)
    outs = [u] + ([c] if c is not None else [])
    g = torch.autograd.grad(outs, [xh], [torch.ones_like(o) for o in outs], create_graph=True)[0]
    return torch.sum(g.view(b, -1).norm(2, dim=1) ** 2)


def discrim_loss(P, prefix, real, fake, conds=None, fake_conds=None, gp_lambda=0.5, alphas=None):
    """D-step loss — `CondGan.discrim_forward` txt2vid/gan/cond_gan.py:34-87 with RSGAN and the
    multi-scale zero-centred GP of txt2vid/gan/losses.py:188-207.
    real/fake: lists of 4 levels. cond path: conds/fake_conds lists (mismatched captions)."""
    if conds is not None:
        real_cc = multiscale_discrim(P, real, conds, prefix)
        real_ic = multiscale_discrim(P, real, fake_conds, prefix)   # reference recomputes the trunk
        fake_cc = multiscale_discrim(P, fake, conds, prefix)
        lu = torch.stack([rsgan_discrim_loss(f[0], r[0]) for f, r in zip(fake_cc, real_cc)]).mean()
        l1 = torch.stack([rsgan_discrim_loss(f[1], r[1]) for f, r in zip(fake_cc, real_cc)]).mean()
        l2 = torch.stack([rsgan_discrim_loss(f[1], r[1]) for f, r in zip(real_ic, real_cc)]).mean()
        l = (lu + (l1 + l2) / 2) / 2.0
    else:
        rp = [r[0] for r in multiscale_discrim(P, real, None, prefix)]
        fp = [f[0] for f in multiscale_discrim(P, fake, None, prefix)]
        l = torch.stack([rsgan_discrim_loss(f, r) for f, r in zip(fp, rp)]).mean()
    if gp_lambda > 0:
        gps = []
        for i in range(len(real)):
            a = None if alphas is None else alphas[i]
            if conds is None:
                gps.append(gp_level(P, prefix, real[i], fake[i], alpha=a))
            else:
                gps.append(gp_level(P, prefix, real[i], fake[i], conds[i], fake_conds[i], alpha=a))
        l = l + gp_lambda * torch.stack(gps).sum()
    return l


def gen_loss(P, prefix, fake, real_pred, conds=None):
    """G-step loss — `CondGan.gen_step` txt2vid/gan/cond_gan.py:90-118 (uncond branch restated with
    `ff[0]`, SURVEY §8a defect 1). `real_pred`: list per level of tensors (uncond) or of
    (u, c, feat) tuples (cond)."""
    fk = multiscale_discrim(P, fake, conds, prefix)
    if conds is None:
        return torch.stack([rsgan_gen_loss(ff[0], rr) for ff, rr in zip(fk, real_pred)]).mean()
    lu = torch.stack([rsgan_gen_loss(ff[0], rr[0]) for ff, rr in zip(fk, real_pred)]).mean()
    lc = torch.stack([rsgan_gen_loss(ff[1], rr[1]) for ff, rr in zip(fk, real_pred)]).mean()
    return (lc + lu) / 2.0


# --------------------------------------------------------------------------------------------
# text encoder — txt2vid/models/txt/basic.py:49-70 (Embedding -> packed 4-layer Bi-LSTM -> last hidden)
# --------------------------------------------------------------------------------------------

def text_encoder_shapes(vocab_size, embed=256, hidden=256, layers=4, prefix='encoder.'):
    """Keys of `Seq2Seq(vocab_size).state_dict()` restricted to the encoder (decoder is the same object)."""
    s = {prefix + 'embed.weight': (vocab_size, embed)}
    h = hidden // 2
    for l in range(layers):
        for suf in ('', '_reverse'):
            s[prefix + 'lstm.weight_ih_l%d%s' % (l, suf)] = (4 * h, embed if l == 0 else hidden)
            s[prefix + 'lstm.weight_hh_l%d%s' % (l, suf)] = (4 * h, h)
            s[prefix + 'lstm.bias_ih_l%d%s' % (l, suf)] = (4 * h,)
            s[prefix + 'lstm.bias_hh_l%d%s' % (l, suf)] = (4 * h,)
    s[prefix + 'to_vocab.weight'] = (vocab_size, hidden)
    s[prefix + 'to_vocab.bias'] = (vocab_size,)
    return s


_LSTM_CACHE = {}


def text_encode(P, tokens, lengths, prefix='encoder.', hidden=256, layers=4):
    """`RecurrentModel.forward(...)[2]`: cat(h_fwd[-1], h_bwd[-1]) of a packed Bi-LSTM."""
    from torch.nn.utils.rnn import pack_padded_sequence
    key = id(P)
    if key not in _LSTM_CACHE:
        emb = P[prefix + 'embed.weight']
        with torch.random.fork_rng():        # the constructor's default init must not consume the global stream
            lstm = torch.nn.LSTM(emb.shape[1], hidden // 2, layers, batch_first=True, bidirectional=True)
        lstm.load_state_dict({k[len(prefix + 'lstm.'):]: v.detach() for k, v in P.items() if k.startswith(prefix + 'lstm.')})
        _LSTM_CACHE[key] = lstm
    lstm = _LSTM_CACHE[key]
    e = F.embedding(tokens, P[prefix + 'embed.weight'])
    _, (hn, _) = lstm(pack_padded_sequence(e, [int(l) for l in lengths], batch_first=True))
    hn = hn.view(layers, 2, -1, hidden // 2)
    return torch.cat((hn[-1, 0], hn[-1, 1]), dim=1)


def text_encode_grad(P, tokens, lengths, prefix='encoder.', hidden=256, layers=4):
    """The same sentence code with the LSTM run on the tensors of `P` themselves (`torch._VF.lstm` on the packed batch, what
    `nn.LSTM.forward` calls), so that gradients reach them: the `--end2end` path (train/gan.py:82-85), where the encoder trains
    with the GAN. Forward values are identical to `text_encode`."""
    from torch.nn.utils.rnn import pack_padded_sequence
    flat = []
    for l in range(layers):
        for suf in ('', '_reverse'):
            flat += [P[prefix + 'lstm.%s_l%d%s' % (n, l, suf)] for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    e = F.embedding(tokens, P[prefix + 'embed.weight'])
    packed = pack_padded_sequence(e, [int(l) for l in lengths], batch_first=True)
    B = int(packed.batch_sizes[0])
    zeros = torch.zeros(layers * 2, B, hidden // 2)
    _, hn, _ = torch._VF.lstm(packed.data, packed.batch_sizes, (zeros, zeros), flat, True, layers, 0.0, True, True)
    hn = hn.view(layers, 2, -1, hidden // 2)
    return torch.cat((hn[-1, 0], hn[-1, 1]), dim=1)


class AdamOnData(object):
    """torch.optim.Adam's arithmetic (single-tensor path, no weight decay / amsgrad) applied through `p.data`: the update does
    not bump the parameters' autograd version counters. That is how the optimiser of the reference's pinned torch 0.4.1 behaved,
    and it is what lets `--end2end` run there: `optD.step()` rewrites the text encoder's weights between the two backward
    passes through ONE encoder graph (trainer.py:240 `retain_graph=... or end2end`), which torch >= 1.x rejects ("modified by an
    inplace operation") when the optimiser updates the parameters themselves."""

    def __init__(self, params, lr, betas, eps=1e-8):
        self.params, self.lr, self.betas, self.eps = list(params), lr, betas, eps
        self.state = {}

    def step(self):
        b1, b2 = self.betas
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), {'step': 0, 'm': torch.zeros_like(p.data), 'v': torch.zeros_like(p.data)})
            st['step'] += 1
            g = p.grad.data
            st['m'].lerp_(g, 1 - b1)
            st['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** st['step'], 1 - b2 ** st['step']
            denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.data.addcdiv_(st['m'], denom, value=-self.lr / bc1)


# --------------------------------------------------------------------------------------------
# the training iteration — txt2vid/gan/trainer.py:190-267
# --------------------------------------------------------------------------------------------

def multiscale_data(x, cond, frame_sizes, subsample_input=True, bts=None):
    """`multiscale_data` — txt2vid/gan/trainer.py:131-165. x: [B,C,T,H,W]."""
    n = len(frame_sizes)
    if n == 1:
        return [x], (None if cond is None else [cond]), []
    xs, conds, used = [], [], []
    for i in range(n):
        if i != n - 1:
            fs = frame_sizes[i]
            xs.append(F.interpolate(x, size=(x.size(2), fs, fs)))
        else:
            xs.append(x)
        if cond is not None:
            conds.append(cond)
        if subsample_input:
            x, bt = subsample(x, None if bts is None else bts[i])
            used.append(bt)
            if cond is not None:
                cond = cond[::2]
    return xs, (conds if conds else None), used


def gen_perm(n):
    """`gen_perm` — txt2vid/util/misc.py:3-8 (numpy global RNG; never returns the identity)."""
    old = np.array(range(n))
    new = np.random.permutation(old)
    while (new == old).all():
        new = np.random.permutation(old)
    return new


class OracleTrainer(object):
    """Holds G / D state dicts (+ Adam) and runs `trainer.train()`'s loop body on the CPU.

    State dict tensors that are parameters are leaf tensors with requires_grad=True; buffers are
    plain tensors. Adam = torch.optim.Adam(lr, betas) exactly as txt2vid/train/gan.py:93-94.
    """

    def __init__(self, PG, PD, d_prefix='single_discrim.', lr=2e-4, betas=(0.5, 0.999),
                 frame_sizes=(8, 16, 32, 64), gp_lambda=0.5, cond_encoder=None, end2end_txt=None):
        """`end2end_txt`: the text encoder's state dict — the `--end2end` mode (train/gan.py:82-85, trainer.py:211-263): its
        parameters join BOTH optimisers, the sentence code keeps its graph, the D backward retains it and the G backward runs
        through it again after `optD.step()` has moved the encoder (see `AdamOnData`)."""
        self.PG, self.PD, self.d_prefix = PG, PD, d_prefix
        self.frame_sizes, self.gp_lambda = list(frame_sizes), gp_lambda
        self.cond_encoder = cond_encoder
        self.g_params = [k for k, v in PG.items() if v.dtype.is_floating_point and 'running_' not in k]
        self.d_params = [k for k, v in PD.items() if v.dtype.is_floating_point]
        for k in self.g_params:
            PG[k].requires_grad_(True)
        for k in self.d_params:
            PD[k].requires_grad_(True)
        self.PT = end2end_txt
        if end2end_txt is None:
            self.optG = torch.optim.Adam([PG[k] for k in self.g_params], lr=lr, betas=betas)
            self.optD = torch.optim.Adam([PD[k] for k in self.d_params], lr=lr, betas=betas)
        else:
            self.t_params = [k for k, v in end2end_txt.items() if 'to_vocab' not in k]        # (the decoder head gets no gradient)
            for k in self.t_params:
                end2end_txt[k].requires_grad_(True)
            txt = [end2end_txt[k] for k in self.t_params]
            self.optG = AdamOnData([PG[k] for k in self.g_params] + txt, lr, betas)
            self.optD = AdamOnData([PD[k] for k in self.d_params] + txt, lr, betas)

    def step_end2end(self, x, tokens, lengths, z=None, latent=256):
        """One `--end2end` iteration (trainer.py:211-263). Returns (lossD, lossG)."""
        cond = text_encode_grad(self.PT, tokens, lengths)                      # NOT detached (trainer.py:213-214)
        B = x.size(0)
        xs, conds, _ = multiscale_data(x, cond, self.frame_sizes)
        if z is None:
            z = torch.randn(B, latent)
        fake = multiscale_gen(self.PG, z, conds[0], training=True)
        self.zero(self.PD, self.d_params)                                       # discrim_step zeroes D and the encoder
        self.zero(self.PT, self.t_params)
        fc0 = conds[0][gen_perm(conds[0].size(0))]
        fake_conds = [fc0[0:c.size(0)] for c in conds]
        lD = discrim_loss(self.PD, self.d_prefix, xs, [f.detach() for f in fake], conds, fake_conds, self.gp_lambda)
        lD.backward(retain_graph=True)                                          # trainer.py:240
        self.optD.step()                                                        # moves D AND the text encoder
        gen_perm(conds[0].size(0))                                              # all_discrim_forward draws a perm it never uses
        real_pred = multiscale_discrim(self.PD, xs, conds, self.d_prefix)       # graph kept: the encoder is reached through it
        self.zero(self.PG, self.g_params)                                       # gen_step zeroes G and the encoder
        self.zero(self.PT, self.t_params)
        lG = gen_loss(self.PD, self.d_prefix, fake, real_pred, conds)
        lG.backward()
        self.optG.step()
        return float(lD.detach()), float(lG.detach())

    def zero(self, P, keys):
        for k in keys:
            P[k].grad = None

    def step(self, x, z=None, cond=None, latent=256):
        """x: [B,C,T,H,W] real batch; returns (lossD, lossG) floats. Random draws follow the
        reference order (SURVEY §7 'RNG parity')."""
        B = x.size(0)
        xs, conds, _ = multiscale_data(x, cond, self.frame_sizes)
        if z is None:
            z = torch.randn(B, latent)
        fake = multiscale_gen(self.PG, z, None if conds is None else conds[0], training=True)
        # ---- D step (cond_gan.py:156-164, trainer.py:231-241)
        self.zero(self.PD, self.d_params)
        fake_conds = None
        if conds is not None:
            fc0 = conds[0][gen_perm(conds[0].size(0))]
            fake_conds = [fc0[0:c.size(0)] for c in conds]
        lD = discrim_loss(self.PD, self.d_prefix, xs, [f.detach() for f in fake], conds, fake_conds,
                          self.gp_lambda)
        lD.backward()
        self.optD.step()
        # ---- real_pred with the updated D (trainer.py:247)
        if conds is not None:
            gen_perm(conds[0].size(0))            # all_discrim_forward draws a perm it never uses
            real_pred = multiscale_discrim(self.PD, xs, conds, self.d_prefix)
        else:
            real_pred = [r[0] for r in multiscale_discrim(self.PD, xs, None, self.d_prefix)]
        # ---- G step (cond_gan.py:90-118, trainer.py:258-263)
        self.zero(self.PG, self.g_params)
        lG = gen_loss(self.PD, self.d_prefix, fake, real_pred, conds)
        lG.backward()
        self.optG.step()
        return float(lD), float(lG)
