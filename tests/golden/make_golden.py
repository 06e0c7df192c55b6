"""Generate golden vectors by running the REAL reference (build container only).

    cd /root/repo && PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (/root/reference, miguelmartin75/txt2vid) is imported read-only with the two runtime
shims of SURVEY.md §8(c); nothing of it is copied. Weights are NOT stored: every state_dict entry
is regenerated from its key by `oracle.tganv2_oracle.recipe_tensor`, here (poured into the reference
modules) and in the tests (poured into the oracle / the HIP modules). The fixtures hold inputs,
outputs, losses, per-key gradient norms and a few small full gradients.
"""
import os
import sys
import random

import numpy as np
import torch
import torch.nn.functional as F
import torch.nn.parallel as TP

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))

TP.data_parallel = lambda m, x, *a, **k: m(x)                                  # shim 1
import txt2vid.gan.losses as RL                                                # noqa: E402
RL.get_labels_for = lambda x, l: torch.full(x.size(), float(l), device=x.device)   # shim 2

from txt2vid.models.layers import (Attention, Attention3d, DownBlock, DownSample, UpBlock,   # noqa: E402
                                   RenderBlock, Subsample)
from txt2vid.models.conv_lstm import ConvLSTM                                   # noqa: E402
from txt2vid.models.resnet3d import Resnet3D                                    # noqa: E402
from txt2vid.models.tganv2.gen import MultiScaleGen as GenU                     # noqa: E402
from txt2vid.models.tganv2.discrim import MultiScaleDiscrim as DisU             # noqa: E402
from txt2vid.models.tganv2_cond.gen import MultiScaleGen as GenC                # noqa: E402
from txt2vid.models.tganv2_cond.discrim import MultiScaleDiscrim as DisC        # noqa: E402
from txt2vid.models.txt.basic import Seq2Seq                                    # noqa: E402
from txt2vid.gan.cond_gan import CondGan                                        # noqa: E402
from txt2vid.gan.losses import MixedGanLoss, RSGANLoss, _gradient_penalty       # noqa: E402
from txt2vid.util.torch.init import init as ref_init                            # noqa: E402

from oracle.tganv2_oracle import recipe_tensor                                  # noqa: E402

torch.set_num_threads(8)


def pour(module, base_seed=0, attn_gamma=0.5):
    """Overwrite every state_dict entry of a reference module by the key recipe."""
    sd = module.state_dict()
    new = {k: recipe_tensor(k, v.shape, base_seed, attn_gamma) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def rnd(seed, *shape):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randn(*shape, generator=g)


def npy(t):
    return t.detach().cpu().numpy()


def grad_norms(module):
    return {k: float(p.grad.norm()) if p.grad is not None else -1.0 for k, p in module.named_parameters()}


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))


def pack_norms(prefix, d, out):
    keys = sorted(d.keys())
    out[prefix + '_keys'] = np.array(keys)
    out[prefix + '_vals'] = np.array([d[k] for k in keys], dtype=np.float64)


# ---------------------------------------------------------------------------------------------
def golden_layers():
    out = {}
    # DownSample incl. odd sizes (layers.py:202-217)
    ds = DownSample()
    for tag, shape in (('a', (2, 3, 4, 6, 6)), ('b', (2, 3, 3, 5, 7)), ('c', (1, 2, 1, 4, 1))):
        x = rnd(1, *shape)
        out['ds_%s_x' % tag] = npy(x)
        out['ds_%s_y' % tag] = npy(ds(x))
    # DownBlock (layers.py:219-243)
    db = pour(DownBlock(in_channels=16, out_channels=32, wide=False))
    x = rnd(2, 2, 16, 4, 6, 6).requires_grad_(True)
    y = db(x)
    gy = rnd(3, *y.shape)
    (y * gy).sum().backward()
    out['db_x'], out['db_y'], out['db_gy'], out['db_gx'] = npy(x), npy(y), npy(gy), npy(x.grad)
    for k, p in db.named_parameters():
        out['db_g_' + k] = npy(p.grad)
    # Attention3d with gamma != 0 (layers.py:39-68) incl. double backward
    at = pour(Attention3d(32))
    x = rnd(4, 2, 32, 2, 4, 4).requires_grad_(True)
    y = at(x)
    gy = rnd(5, *y.shape)
    gx, = torch.autograd.grad((y * gy).sum(), x, create_graph=True)
    r = (gx ** 2).sum()
    r.backward()
    out['at3_x'], out['at3_y'], out['at3_gy'], out['at3_gx'] = npy(x), npy(y), npy(gy), npy(gx)
    out['at3_r'] = np.float64(r.item())
    out['at3_ggx'] = npy(x.grad)
    for k, p in at.named_parameters():
        out['at3_gg_' + k] = npy(p.grad)
    # Attention (2-D) (layers.py:10-36)
    a2 = pour(Attention(32))
    x = rnd(6, 3, 32, 8, 8).requires_grad_(True)
    y = a2(x)
    gy = rnd(7, *y.shape)
    (y * gy).sum().backward()
    out['at2_x'], out['at2_y'], out['at2_gy'], out['at2_gx'] = npy(x), npy(y), npy(gy), npy(x.grad)
    for k, p in a2.named_parameters():
        out['at2_g_' + k] = npy(p.grad)
    # UpBlock (train mode BN) (layers.py:152-195)
    for tag, cin, cout in (('ub', 16, 8), ('ub_same', 8, 8)):
        ub = pour(UpBlock(in_channels=cin, out_channels=cout))
        ub.train()
        x = rnd(8, 4, cin, 4, 4).requires_grad_(True)
        y = ub(x)
        gy = rnd(9, *y.shape)
        (y * gy).sum().backward()
        out[tag + '_x'], out[tag + '_y'], out[tag + '_gy'], out[tag + '_gx'] = npy(x), npy(y), npy(gy), npy(x.grad)
        for k, p in ub.named_parameters():
            out[tag + '_g_' + k] = npy(p.grad)
        for k, v in ub.state_dict().items():
            if 'running' in k:
                out[tag + '_buf_' + k] = npy(v)
    # RenderBlock (layers.py:245-259)
    rb = pour(RenderBlock(in_channels=8, out_channels=3))
    rb.train()
    x = rnd(10, 4, 8, 8, 8).requires_grad_(True)
    y = rb(x)
    gy = rnd(11, *y.shape)
    (y * gy).sum().backward()
    out['rb_x'], out['rb_y'], out['rb_gy'], out['rb_gx'] = npy(x), npy(y), npy(gy), npy(x.grad)
    for k, p in rb.named_parameters():
        out['rb_g_' + k] = npy(p.grad)
    # ConvLSTM, h=w=1 and 2 (conv_lstm.py:57-97); keys get the generator's prefix
    for tag, hw in (('cl1', 1), ('cl2', 2)):
        cl = ConvLSTM(input_channels=8, hidden_channels=[8], kernel_size=3, step=5, effective_step=range(5))
        sd = {k: recipe_tensor('clstm.' + k, v.shape) for k, v in cl.state_dict().items()}
        cl.load_state_dict(sd)
        x = rnd(12, 3, 8, hw, hw).requires_grad_(True)
        ys, _ = cl(x)
        y = torch.stack(ys)
        gy = rnd(13, *y.shape)
        (y * gy).sum().backward()
        out[tag + '_x'], out[tag + '_y'], out[tag + '_gy'], out[tag + '_gx'] = npy(x), npy(y), npy(gy), npy(x.grad)
        for k, p in cl.named_parameters():
            out[tag + '_g_' + k] = npy(p.grad)
    # Subsample (layers.py:98-111)
    x = rnd(14, 5, 2, 6, 3, 3)
    ss = Subsample()
    out['ss_x'] = npy(x)
    out['ss_y0'] = npy(ss(x, bt=0)[0])
    out['ss_y1'] = npy(ss(x, bt=1)[0])
    save('layers', **out)


# ---------------------------------------------------------------------------------------------
def golden_resnet3d():
    """Resnet3D heads, first-order grads and the GP double backward (resnet3d.py:38-57,
    losses.py:135-186) on a small video, full-size channels."""
    out = {}
    for tag, cond_dim in (('u', 0), ('c', 24)):
        net = pour(Resnet3D(num_channels=1, cond_dim=cond_dim))
        x = rnd(20, 2, 1, 4, 16, 16).requires_grad_(True)
        cond = rnd(21, 2, cond_dim) if cond_dim else None
        u, c, feat = net(x, cond=cond)
        loss = (u * rnd(22, 2, 1)).sum() + (feat * rnd(23, 2, 1024)).sum() * 1e-2
        if c is not None:
            loss = loss + (c * rnd(24, 2, 1)).sum()
        loss.backward()
        out[tag + '_x'] = npy(x)
        if cond is not None:
            out[tag + '_cond'] = npy(cond)
            out[tag + '_c'] = npy(c)
        out[tag + '_u'], out[tag + '_feat'], out[tag + '_gx'] = npy(u), npy(feat), npy(x.grad)
        pack_norms(tag + '_gn', grad_norms(net), out)
        out[tag + '_g_fc_uncond.weight'] = npy(net.fc_uncond.weight.grad)
        out[tag + '_g_res_block.inner_module.0.weight'] = npy(net.res_block.inner_module[0].weight.grad)
        # gradient penalty: zero-centred, sum-combined (losses.py:203)
        net.zero_grad()
        xr, xf = rnd(25, 2, 1, 4, 16, 16), rnd(26, 2, 1, 4, 16, 16)
        cr = rnd(27, 2, cond_dim) if cond_dim else None
        cf = rnd(28, 2, cond_dim) if cond_dim else None
        torch.manual_seed(77)
        gp = _gradient_penalty(net, real_x=xr, fake_x=xf, real_cond=cr, fake_cond=cf, zero_center=True,
                               combine=torch.sum)
        gp.backward()
        out[tag + '_gp_xr'], out[tag + '_gp_xf'] = npy(xr), npy(xf)
        if cond_dim:
            out[tag + '_gp_cr'], out[tag + '_gp_cf'] = npy(cr), npy(cf)
        out[tag + '_gp'] = np.float64(gp.item())
        pack_norms(tag + '_gp_gn', grad_norms(net), out)
        out[tag + '_gp_g_down.1.gamma'] = npy(net.down[1].gamma.grad)
        out[tag + '_gp_g_down.1.theta.weight'] = npy(net.down[1].theta.weight.grad)
        out[tag + '_gp_g_res_block.inner_module.0.weight'] = npy(net.res_block.inner_module[0].weight.grad)
        out[tag + '_gp_g_fc_uncond.weight'] = npy(net.fc_uncond.weight.grad)
    save('resnet3d', **out)


# ---------------------------------------------------------------------------------------------
def golden_gen():
    """MultiScaleGen train-mode pyramid + eval-mode video (tganv2/gen.py:62-119,
    tganv2_cond/gen.py:64-124)."""
    out = {}
    for tag, cls, cond_dim in (('u', GenU, 0), ('c', GenC, 16)):
        g = pour(cls(width=64, height=64, num_channels=1, cond_dim=cond_dim))
        g.train()
        B = 8
        z = rnd(30, B, 256)
        cond = rnd(31, B, cond_dim) if cond_dim else None
        torch.manual_seed(5)
        bts = [int(torch.randint(2, (1,))) for _ in range(3)]
        torch.manual_seed(5)
        fake = g(z, cond=cond)
        loss = sum((f * rnd(40 + i, *f.shape)).sum() for i, f in enumerate(fake))
        loss.backward()
        out[tag + '_z'] = npy(z)
        if cond is not None:
            out[tag + '_cond'] = npy(cond)
        out[tag + '_bts'] = np.array(bts)
        for i, f in enumerate(fake):
            out[tag + '_fake%d' % i] = npy(f)
        pack_norms(tag + '_gn', grad_norms(g), out)
        out[tag + '_g_fc.bias'] = npy(g.fc.bias.grad)
        out[tag + '_g_render_blocks.3.conv.weight'] = npy(g.render_blocks[3].conv.weight.grad)
        for k, v in g.state_dict().items():
            if k.endswith('render_blocks.3.bn.running_mean') or k.endswith('render_blocks.3.bn.running_var') \
                    or k.endswith('up0.main.inner_module.0.running_var'):
                out[tag + '_buf_' + k] = npy(v)
        g.eval()
        with torch.no_grad():
            vid = g(z[:2], cond=None if cond is None else cond[:2])
        assert len(vid) == 1
        out[tag + '_eval'] = npy(vid[0])
    save('gen', **out)


# ---------------------------------------------------------------------------------------------
def golden_steps():
    """Losses of training iterations 0..2, reference loop body restated from trainer.py:199-267
    (SURVEY Appendix B), uncond TGANv2 64x64x1, B=4, recipe weights, RSGAN + GP 0.5, Adam 2e-4."""
    out = {}
    B = 4
    g = pour(GenU(width=64, height=64, num_channels=1))
    d = pour(DisU(num_channels=1))
    seed = 100                       # seeded AFTER construction: module ctors consume the global RNG
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    g.train()
    d.train()
    gan = CondGan(gen=g, discrims=[d], discrim_names=['video'])
    losses = MixedGanLoss(g_loss=RSGANLoss(), d_loss=RSGANLoss())
    optD = torch.optim.Adam([{'params': d.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam([{'params': g.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    ss = Subsample()
    fs = [8, 16, 32, 64]
    lD_all, lG_all = [], []
    for it in range(3):
        x = (torch.rand(B, 16, 1, 64, 64) * 2 - 1).permute(0, 2, 1, 3, 4)
        xs = []
        for i in range(4):
            xs.append(F.interpolate(x, size=(x.size(2), fs[i], fs[i])) if i != 3 else x)
            x, _ = ss(x)
        z = torch.randn(B, g.latent_size)
        fake = gan(z, cond=None)
        lD = gan.discrim_step(real=xs, fake=[f.detach() for f in fake], cond=None,
                              loss=losses.discrim_loss, gp_lambda=0.5)
        lD.backward()
        if it == 0:
            pack_norms('it0_D_gn', grad_norms(d), out)
        optD.step()
        _, _, real_pred = gan.all_discrim_forward(real=xs, cond=None, fake=None, loss=None)
        g.zero_grad()
        fc = d(x=fake, cond=None, xbar=None)
        lG = torch.stack([losses.gen_loss(fake=ff[0], real=rr) for ff, rr in zip(fc, real_pred[0])]).mean()
        lG.backward()
        if it == 0:
            pack_norms('it0_G_gn', grad_norms(g), out)
        optG.step()
        lD_all.append(float(lD))
        lG_all.append(float(lG))
        print('uncond it', it, float(lD), float(lG))
    out['lossD'] = np.array(lD_all, dtype=np.float64)
    out['lossG'] = np.array(lG_all, dtype=np.float64)
    save('steps_uncond', **out)


def golden_steps_cond():
    """Text-conditioned TGANv2 (Bi-LSTM cond 256-d, non-local blocks on), B=4: the reference's own
    `discrim_step` / `all_discrim_forward` / `gen_step` (cond_gan.py:90-164) for 3 iterations."""
    out = {}
    B, V = 4, 21
    g = pour(GenC(width=64, height=64, num_channels=1, cond_dim=256))
    d = pour(DisC(num_channels=1, cond_dim=256))
    txt = Seq2Seq(vocab_size=V)          # encoder and decoder are ONE module: both key prefixes alias the same tensors
    txt.load_state_dict({k: recipe_tensor('encoder.' + k.split('.', 1)[1], v.shape) for k, v in txt.state_dict().items()})
    g.train()
    d.train()
    seed = 100
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    gan = CondGan(gen=g, discrims=[d], cond_encoder=txt, discrim_names=['video'])
    losses = MixedGanLoss(g_loss=RSGANLoss(), d_loss=RSGANLoss())
    optD = torch.optim.Adam([{'params': d.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam([{'params': g.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    ss = Subsample()
    fs = [8, 16, 32, 64]
    tg = torch.Generator()
    tg.manual_seed(1234)
    tokens = torch.randint(4, V, (B, 8), generator=tg)
    tokens[:, 0], tokens[:, -1] = 1, 2
    lengths = [8] * B
    out['tokens'] = tokens.numpy()
    lD_all, lG_all = [], []
    for it in range(3):
        x = (torch.rand(B, 16, 1, 64, 64) * 2 - 1).permute(0, 2, 1, 3, 4)
        _, _, cond = gan.cond_encoder.encode(tokens, lengths)
        cond = cond.detach()
        if it == 0:
            out['cond0'] = npy(cond)
        xs, conds = [], []
        for i in range(4):
            xs.append(F.interpolate(x, size=(x.size(2), fs[i], fs[i])) if i != 3 else x)
            conds.append(cond)
            x, _ = ss(x)
            cond = cond[::2]
        z = torch.randn(B, g.latent_size)
        fake = gan(z, cond=conds[0])
        lD = gan.discrim_step(real=xs, fake=[f.detach() for f in fake], cond=conds, loss=losses.discrim_loss, gp_lambda=0.5)
        lD.backward()
        if it == 0:
            pack_norms('it0_D_gn', grad_norms(d), out)
        optD.step()
        _, _, real_pred = gan.all_discrim_forward(real=xs, cond=conds, fake=None, loss=None)
        lG = gan.gen_step(fake=fake, real_pred=real_pred, cond=conds, loss=losses.gen_loss)
        lG.backward()
        if it == 0:
            pack_norms('it0_G_gn', grad_norms(g), out)
        optG.step()
        lD_all.append(float(lD))
        lG_all.append(float(lG))
        print('cond it', it, float(lD), float(lG))
    out['lossD'] = np.array(lD_all, dtype=np.float64)
    out['lossG'] = np.array(lG_all, dtype=np.float64)
    save('steps_cond', **out)


def golden_init():
    """Checksums of `init(model, 'xavier')` after seeding 100 and constructing G then D
    (train/setup.py:7-14, train/gan.py:60-70, util/torch/init.py:4-39)."""
    out = {}
    random.seed(100)
    np.random.seed(100)
    torch.manual_seed(100)
    g = GenU(width=64, height=64, num_channels=1)
    d = DisU(num_channels=1)
    ref_init(g, 'xavier')
    ref_init(d, 'xavier')
    for tag, m in (('G', g), ('D', d)):
        sd = m.state_dict()
        keys = sorted(k for k, v in sd.items() if v.dtype.is_floating_point)
        out[tag + '_keys'] = np.array(keys)
        out[tag + '_sum'] = np.array([float(sd[k].double().sum()) for k in keys])
        out[tag + '_abs'] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    save('init_xavier', **out)


def golden_losses():
    """The loss zoo (gan/losses.py:19-133) on [12,1] logits: both sides' values and their gradients w.r.t. the real and
    the fake logits. RaSGANLoss as written reads attributes it never sets (`fake_labels` :95); they are SET on the
    instance here (0 / 1, the values its constructor stores under the singular names), nothing is patched."""
    out = {'real': npy(rnd(71, 12, 1) * 2.0), 'fake': npy(rnd(72, 12, 1) * 2.0 + 0.3)}
    ra = RL.RaSGANLoss()
    ra.fake_labels, ra.real_labels = ra.fake_label, ra.real_label
    zoo = {'vanilla': RL.VanillaGanLoss(), 'hinge': RL.HingeGanLoss(), 'hinge3': RL.HingeGanLoss(margin=3.0),
           'wasserstein': RL.WassersteinGanLoss(), 'rasgan': ra, 'ralsgan': RL.RaLSGANLoss(), 'rsgan': RL.RSGANLoss()}
    for name, obj in zoo.items():
        for side, fn in ((0, obj.discrim_loss), (1, obj.gen_loss)):
            r = torch.tensor(out['real'], requires_grad=True)
            f = torch.tensor(out['fake'], requires_grad=True)
            loss = fn(fake=f, real=r)
            gr, gf = torch.autograd.grad(loss, [r, f], allow_unused=True)
            out['%s.%d.loss' % (name, side)] = npy(loss)
            out['%s.%d.g_real' % (name, side)] = npy(gr if gr is not None else torch.zeros_like(r))
            out['%s.%d.g_fake' % (name, side)] = npy(gf if gf is not None else torch.zeros_like(f))
    save('losses', **out)


def golden_txt_pretrain():
    """One iteration of the text auto-encoder pre-training exactly as train/txt.py:160-178 runs it — `encode`, `decode` from the
    encoder's state (teacher-forced and greedy), `nn.CrossEntropyLoss` on `decoded.permute(0, 2, 1)`, backward — on a ragged
    batch (lengths 7,6,6,4,2), V=37, recipe weights (encoder and decoder are ONE module)."""
    out = {}
    V, B = 37, 5
    lengths = [7, 6, 6, 4, 2]
    txt = Seq2Seq(vocab_size=V)
    # matrices x4: with the plain xavier-scale recipe the logits are bias-dominated and greedy decoding emits one constant symbol
    txt.load_state_dict({k: recipe_tensor('encoder.' + k.split('.', 1)[1], v.shape) * (4.0 if v.dim() >= 2 else 1.0)
                         for k, v in txt.state_dict().items()})
    tg = torch.Generator()
    tg.manual_seed(4321)
    sent = torch.zeros(B, lengths[0], dtype=torch.long)
    for b, n in enumerate(lengths):
        sent[b, :n] = torch.randint(4, V, (n,), generator=tg)
        sent[b, 0], sent[b, n - 1] = 1, 2
    out['tokens'] = sent.numpy()
    out['lengths'] = np.array(lengths)
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    criteria = torch.nn.CrossEntropyLoss()
    for tag, tf in (('tf', True), ('greedy', False)):
        txt.zero_grad()
        enc_out, hs, hn = txt.encode(sent, lengths=lengths)
        packed = pack_padded_sequence(sent, lengths, batch_first=True)
        targets, _ = pad_packed_sequence(packed, batch_first=True, total_length=lengths[0])
        decoded, symbols = txt.decode(true_inputs=sent, initial_hidden=hs, max_seq_len=lengths[0], teacher_force=tf)
        loss = criteria(decoded.permute(0, 2, 1), targets)
        loss.backward()
        out[tag + '_loss'] = np.float64(loss.item())
        out[tag + '_sum_loss'] = np.float64(torch.nn.CrossEntropyLoss(reduction='sum')(decoded.permute(0, 2, 1), targets).item())
        out[tag + '_decoded'] = npy(decoded)
        out[tag + '_symbols'] = symbols.numpy()
        out[tag + '_hn'] = npy(hn)
        out[tag + '_enc_out'] = npy(enc_out)
        out[tag + '_h_n'] = npy(hs[0])
        out[tag + '_c_n'] = npy(hs[1])
        pack_norms(tag + '_gn', {k: v for k, v in grad_norms(txt).items() if k.startswith('encoder.')}, out)
        named = dict(txt.named_parameters())
        for k in ('encoder.embed.weight', 'encoder.to_vocab.bias', 'encoder.lstm.bias_hh_l0', 'encoder.lstm.bias_ih_l3_reverse',
                  'encoder.lstm.weight_hh_l1_reverse'):
            out[tag + '_g_' + k] = npy(named[k].grad)
    save('txt_pretrain', **out)


# ---------------------------------------------------------------------------------------------
def golden_host():
    """Which ATen CPU kernels produced the fixtures: torch version, CPU ISA level and thread count. tests/test_oracle_golden.py
    asserts the tight (<= 1e-6) oracle-vs-reference bound when the host runs the same kernels and the looser cross-platform
    bound otherwise (a different ISA changes oneDNN's summation order)."""
    import json
    info = {'torch': torch.__version__, 'cpu_capability': torch.backends.cpu.get_cpu_capability(), 'threads': torch.get_num_threads()}
    with open(os.path.join(HERE, 'host.json'), 'w') as f:
        json.dump(info, f)
    print('wrote host.json', info)


def _compact(obj):
    """Same nested structure with every tensor replaced by a stride-0 view of ONE element (its mean): shape, dtype and key
    layout survive `torch.save`, the bytes do not (the generator alone is 345 MB; fixtures must stay small)."""
    if isinstance(obj, torch.Tensor):
        if obj.numel() == 0 or not obj.dtype.is_floating_point:
            return obj.detach().clone()
        v = obj.detach().double().mean().to(obj.dtype).reshape(1)
        return v.expand(obj.numel()).view(obj.shape) if obj.dim() else v.reshape(())
    if isinstance(obj, dict):
        return type(obj)((k, _compact(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_compact(v) for v in obj)
    return obj


def golden_checkpoint():
    """The dict `trainer.train()` hands to `torch.save` (trainer.py:269-279): `{'optG', 'optD'}` + `CondGan.save_dict()`
    (cond_gan.py:186-196), built by the REFERENCE's classes after one real optimiser step each (so both Adam states
    exist), for the text-conditioned model (`single_discrim.module.*` keys from its nn.DataParallel wrapper, 'cond' entry) and the
    unconditional one (`single_discrim.*`). Tensor VALUES are compacted to one element per tensor (`_compact`): what is pinned
    is the file format — key names, nesting, shapes, dtypes, optimiser param-group layout."""
    for tag, G, D, cond_dim in (('uncond', GenU, DisU, 0), ('cond', GenC, DisC, 256)):
        g = pour(G(width=64, height=64, num_channels=1, **({'cond_dim': cond_dim} if cond_dim else {})))
        d = pour(D(num_channels=1, **({'cond_dim': cond_dim} if cond_dim else {})))
        txt = None
        if cond_dim:
            txt = Seq2Seq(vocab_size=21)
            txt.load_state_dict({k: recipe_tensor('encoder.' + k.split('.', 1)[1], v.shape) for k, v in txt.state_dict().items()})
        gan = CondGan(gen=g, discrims=[d], cond_encoder=txt, discrim_names=['video'])
        optD = torch.optim.Adam([{'params': d.parameters()}], lr=2e-4, betas=(0.5, 0.999))
        optG = torch.optim.Adam([{'params': g.parameters()}], lr=2e-4, betas=(0.5, 0.999))
        for m, opt in ((g, optG), (d, optD)):                     # one step on synthetic gradients: populates the Adam state
            for i, p in enumerate(m.parameters()):
                p.grad = torch.full_like(p, 1e-3 * (1 + i % 7))
            opt.step()
        to_save = {'optG': optG.state_dict(), 'optD': optD.state_dict()}
        to_save.update(gan.save_dict())
        path = os.path.join(HERE, 'ref_checkpoint_%s.pt' % tag)
        torch.save(_compact(to_save), path)
        print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0), sorted(to_save.keys()))


def golden_seq2seq_pickle():
    """`train/txt.py:185` pickles the WHOLE module: `torch.save({'optim': optimizer, 'txt': seq2seq}, path)`; `train/gan.py
    --sent_weights` unpickles it (train/gan.py:36-43). Written here by the reference's own classes (class path
    `txt2vid.models.txt.basic.Seq2Seq`; its instances have no attribute this build adds in __init__), parameters compacted."""
    txt = Seq2Seq(vocab_size=21)
    opt = torch.optim.Adam(txt.parameters(), lr=1e-3)
    for p in txt.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    for p in txt.parameters():
        p.grad = None
    with torch.no_grad():
        for mod in txt.modules():
            for name, p in list(mod._parameters.items()):
                if p is not None:
                    p.data = _compact(p.data)
            if hasattr(mod, '_flat_weights'):
                mod._flat_weights = [getattr(mod, n) for n in mod._flat_weights_names]
    for st in opt.state.values():
        for k, v in list(st.items()):
            st[k] = _compact(v)
    path = os.path.join(HERE, 'ref_seq2seq.pt')
    torch.save({'optim': opt, 'txt': txt}, path)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))


def golden_data():
    """The input-pipeline contract (txt2vid/data/__init__.py): `Vocab` / `build_vocab` / `tokenize` / `to_words`, `pick_frames`,
    `collate_fn`, and `Dataset.__getitem__` on two tiny frame folders (JPEG bytes stored in the fixture). The reference module
    hard-imports cv2, torchvision and nvidia.dali at the top (data/__init__.py:6,9-10,16-18), none of which exist here and none of
    which the functions above touch: EMPTY placeholder modules are registered for those names so that the import statement
    succeeds (runtime shims like SURVEY §8c's two; nothing of those libraries is emulated). The per-frame transform is passed in
    by the caller (the reference's own `default_transform` is torchvision code)."""
    import io
    import pickle
    import tempfile
    import types
    from PIL import Image
    for name in ('cv2', 'torchvision', 'torchvision.transforms', 'torchvision.transforms.functional', 'nvidia', 'nvidia.dali',
                 'nvidia.dali.pipeline', 'nvidia.dali.ops', 'nvidia.dali.types'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision.transforms'].functional = sys.modules['torchvision.transforms.functional']
    sys.modules['nvidia'].dali = sys.modules['nvidia.dali']
    sys.modules['nvidia.dali'].ops = sys.modules['nvidia.dali.ops']
    sys.modules['nvidia.dali'].types = sys.modules['nvidia.dali.types']
    sys.modules['nvidia.dali.pipeline'].Pipeline = object
    import txt2vid.data as RD
    out = {}
    sentences = ['digit 3 is left and right.', 'A man is playing a guitar', 'the cat. sat on The mat', 'digit 7 is top and bottom.',
                 'someone slices an onion. quickly']
    vocab = RD.build_vocab(sentences)
    words = [vocab.idx2word[i] for i in range(len(vocab))]
    out['vocab_words'] = np.array(words)
    out['sentences'] = np.array(sentences)
    probes = sentences + ['an unseen word appears. here', 'digit 3 is']
    out['probes'] = np.array(probes)
    toks = [[vocab(t) for t in vocab.tokenize(s)] for s in probes]
    out['probe_lens'] = np.array([len(t) for t in toks])
    out['probe_tokens'] = np.array(sum(toks, []))
    out['probe_words'] = np.array([vocab.to_words(t) for t in toks])
    # pick_frames, deterministic branch
    for n in (16, 17, 40, 64):
        out['pick_%d' % n] = np.array(RD.pick_frames(list(range(100, 100 + n)), random=False, num_frames=16))
    # Dataset on two frame folders (+ one listed video that is missing on disk)
    rs = np.random.RandomState(7)

    def transform(img):                                           # caller-supplied: float32 [C,h,w] in [-1,1]
        a = np.asarray(img.convert('L'), dtype=np.float32) / 255.0
        return torch.from_numpy((a[None] - 0.5) / 0.5)
    with tempfile.TemporaryDirectory() as tmp:
        captions = {'vidA': ['digit 3 is left and right.', 'the cat. sat on The mat'], 'vidB': ['A man is playing a guitar'],
                    'vidMissing': ['never read']}
        nframes = {'vidA': 20, 'vidB': 33}
        for vid, n in nframes.items():
            os.makedirs(os.path.join(tmp, vid))
            for i in range(n):
                img = Image.fromarray(rs.randint(0, 256, size=(12, 12, 3)).astype(np.uint8))
                buf = io.BytesIO()
                img.save(buf, format='JPEG', quality=90)
                data = buf.getvalue()
                with open(os.path.join(tmp, vid, '%d.jpg' % (i * 3)), 'wb') as f:      # frame ids 0, 3, 6, ...: sorted numerically
                    f.write(data)
                out['jpeg_%s_%d' % (vid, i * 3)] = np.frombuffer(data, dtype=np.uint8)
            with open(os.path.join(tmp, vid, 'notes.txt'), 'w') as f:                   # non-frame files are skipped
                f.write('x')
        cap_path = os.path.join(tmp, 'captions.pkl')
        with open(cap_path, 'wb') as f:
            pickle.dump(captions, f)
        out['captions_pickle'] = np.frombuffer(pickle.dumps(captions, protocol=2), dtype=np.uint8)
        ds = RD.Dataset(video_dir=tmp, vocab=vocab, captions=cap_path, transform=transform)
        out['ds_len'] = np.int64(len(ds))
        out['ds_missing'] = np.int64(ds.missing)
        out['ds_video_ids'] = np.array([str(v) for v in ds.video_ids])
        items = [ds[i] for i in range(len(ds))]
        for i, (frames, cap) in enumerate(items):
            out['ds_frames_%d' % i] = npy(frames)
            out['ds_caption_%d' % i] = npy(cap)
            out['ds_caption_dtype_%d' % i] = np.array(str(cap.dtype))
        vids, targets, lengths = RD.collate_fn(list(items))
        out['collate_vids'], out['collate_targets'], out['collate_lengths'] = npy(vids), targets.numpy(), np.array(lengths)
        out['collate_targets_dtype'] = np.array(str(targets.dtype))
    save('data_contract', **out)


def golden_trajectory(n_steps=100):
    """the CPU oracle and the imported reference side by side for `n_steps` FREE-RUNNING iterations in this
    container (uncond TGANv2 64x64x1, B=4, recipe weights, seeds 100): per-step |delta lossD|, |delta lossG| (expected exactly 0:
    same ATen kernels in the same order) and the reference's own loss curve. Only these numbers are committed
    (tests/golden/trajectory_%d.json); the per-step states are 345 MB each."""
    import json
    import time
    from oracle import tganv2_oracle as O
    B = 4
    g = pour(GenU(width=64, height=64, num_channels=1))
    d = pour(DisU(num_channels=1))
    g.train()
    d.train()
    gan = CondGan(gen=g, discrims=[d], discrim_names=['video'])
    losses = MixedGanLoss(g_loss=RSGANLoss(), d_loss=RSGANLoss())
    optD = torch.optim.Adam([{'params': d.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam([{'params': g.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    tr = O.OracleTrainer(O.recipe_state(O.gen_shapes(num_channels=1)), O.recipe_state(O.resnet3d_shapes('single_discrim.', 1, 64, 0)))
    ss = Subsample()
    fs = [8, 16, 32, 64]
    rec = {'lossD_ref': [], 'lossG_ref': [], 'dD': [], 'dG': [], 'max_param_delta': []}
    seed = 100
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    t0 = time.time()
    for it in range(n_steps):
        x0 = (torch.rand(B, 16, 1, 64, 64) * 2 - 1).permute(0, 2, 1, 3, 4)
        state = (torch.get_rng_state(), np.random.get_state(), random.getstate())
        x = x0
        xs = []
        for i in range(4):
            xs.append(F.interpolate(x, size=(x.size(2), fs[i], fs[i])) if i != 3 else x)
            x, _ = ss(x)
        z = torch.randn(B, g.latent_size)
        fake = gan(z, cond=None)
        lD = gan.discrim_step(real=xs, fake=[f.detach() for f in fake], cond=None, loss=losses.discrim_loss, gp_lambda=0.5)
        lD.backward()
        optD.step()
        _, _, real_pred = gan.all_discrim_forward(real=xs, cond=None, fake=None, loss=None)
        g.zero_grad()
        fc = d(x=fake, cond=None, xbar=None)
        lG = torch.stack([losses.gen_loss(fake=ff[0], real=rr) for ff, rr in zip(fc, real_pred[0])]).mean()
        lG.backward()
        optG.step()
        after = (torch.get_rng_state(), np.random.get_state(), random.getstate())
        torch.set_rng_state(state[0])
        np.random.set_state(state[1])
        random.setstate(state[2])
        lDo, lGo = tr.step(x0)
        torch.set_rng_state(after[0])
        np.random.set_state(after[1])
        random.setstate(after[2])
        pd = 0.0
        if it % 10 == 9 or it == n_steps - 1:                      # parameters of both models, every 10th step
            sd_g, sd_d = g.state_dict(), d.state_dict()
            pd = max(max(float((sd_g[k] - tr.PG[k].detach()).abs().max()) for k in tr.g_params),
                     max(float((sd_d[k] - tr.PD[k].detach()).abs().max()) for k in tr.d_params))
        rec['lossD_ref'].append(float(lD))
        rec['lossG_ref'].append(float(lG))
        rec['dD'].append(abs(float(lD) - lDo))
        rec['dG'].append(abs(float(lG) - lGo))
        rec['max_param_delta'].append(pd)
        print('it %3d  ref lossD %.7f lossG %.7f | oracle delta %.1e %.1e | param delta %.1e | %.0f s' %
              (it, float(lD), float(lG), rec['dD'][-1], rec['dG'][-1], pd, time.time() - t0), flush=True)
    rec['summary'] = {'steps': n_steps, 'max_dD': max(rec['dD']), 'max_dG': max(rec['dG']), 'max_param_delta': max(rec['max_param_delta']),
                      'torch': torch.__version__, 'threads': torch.get_num_threads(), 'batch': B,
                      'protocol': 'free-running, both consume identical x / z / subsample phases / GP alphas'}
    with open(os.path.join(HERE, 'trajectory_%d.json' % n_steps), 'w') as f:
        json.dump(rec, f)
    print('summary', rec['summary'])


if __name__ == '__main__':
    which = sys.argv[1:] or ['layers', 'resnet3d', 'gen', 'steps', 'init', 'steps_cond', 'losses', 'txt_pretrain']
    if 'losses' in which:
        golden_losses()
    if 'txt_pretrain' in which:
        golden_txt_pretrain()
    if 'layers' in which:
        golden_layers()
    if 'resnet3d' in which:
        golden_resnet3d()
    if 'gen' in which:
        golden_gen()
    if 'init' in which:
        golden_init()
    if 'steps' in which:
        golden_steps()
    if 'steps_cond' in which:
        golden_steps_cond()
    if 'host' in which:
        golden_host()
    if 'checkpoint' in which:
        golden_checkpoint()
    if 'seq2seq_pickle' in which:
        golden_seq2seq_pickle()
    if 'data' in which:
        golden_data()
    if 'trajectory' in which:                                       # python make_golden.py trajectory [N]   (~15 s per step)
        n = [int(a) for a in which if a.isdigit()]
        golden_trajectory(n[0] if n else 100)
