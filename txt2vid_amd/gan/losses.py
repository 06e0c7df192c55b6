"""GAN losses of the hot path — same classes / call signatures as txt2vid/gan/losses.py.

RSGANLoss (losses.py:74-85) and the gradient penalty (losses.py:135-209) are the north-star path; the rest of
the zoo (Vanilla / Hinge / Wasserstein / RaSGAN / RaLSGAN) shares one HIP loss-head kernel.
"""
import torch

from .. import functional as TF


def get_labels_for(x, label):
    return torch.full(x.size(), float(label), device=x.device)


class MixedGanLoss(object):
    """Separate G / D loss objects — losses.py:8-17."""

    def __init__(self, g_loss=None, d_loss=None):
        self.g_loss, self.d_loss = g_loss, d_loss

    def discrim_loss(self, fake=None, real=None):
        return self.d_loss.discrim_loss(fake=fake, real=real)

    def gen_loss(self, fake=None, real=None):
        return self.g_loss.gen_loss(fake=fake, real=real)

    def __getattr__(self, name):
        # the all-levels-in-one-launch forms (cond_gan._mean_over_levels) exist exactly when the wrapped loss offers them
        if name == 'discrim_loss_levels':
            return getattr(self.__dict__['d_loss'], name)
        if name == 'gen_loss_levels':
            return getattr(self.__dict__['g_loss'], name)
        raise AttributeError(name)


class RSGANLoss(object):
    """Relativistic standard GAN: BCEWithLogits(real - fake, 1) / BCEWithLogits(fake - real, 1)."""

    def __init__(self, bce_loss=True):
        if not bce_loss:
            raise NotImplementedError('only the BCE form is used by the reference scripts')

    def discrim_loss(self, fake=None, real=None):
        return TF.rsgan(real, fake)

    def gen_loss(self, fake=None, real=None):
        return TF.rsgan(fake, real)

    # all pyramid levels at once (cond_gan._mean_over_levels): mean over levels of the per-level loss, one launch
    def discrim_loss_levels(self, fakes=None, reals=None):
        return TF.rsgan_mean_levels(reals, fakes)

    def gen_loss_levels(self, fakes=None, reals=None):
        return TF.rsgan_mean_levels(fakes, reals)


class _ZooLoss(object):
    """A loss of the zoo = one `t2v_gan_loss` launch per call (and one for its gradient)."""
    kind, margin = None, 0.0

    def discrim_loss(self, fake=None, real=None):
        return TF.gan_loss(self.kind, 0, real, fake, self.margin)

    def gen_loss(self, fake=None, real=None):
        return TF.gan_loss(self.kind, 1, real, fake, self.margin)


class VanillaGanLoss(_ZooLoss):
    """losses.py:19-46. `LabelledGanLoss.__init__` stores the labels crossed (:27-28), so the reference's
    D is trained towards fake -> 1, real -> 0 and G towards fake -> 0; consistent, and kept as is."""
    kind = 'vanilla'

    def __init__(self, bce_loss=True, reduction='mean'):
        if not bce_loss or reduction != 'mean':
            raise NotImplementedError('the reference scripts only use BCE-with-logits, mean reduction')


class HingeGanLoss(_ZooLoss):
    """losses.py:48-52: nn.HingeEmbeddingLoss(margin) with labels fake -> 1, real -> -1 (crossed as above)."""
    kind = 'hinge'

    def __init__(self, margin=2.0):
        self.margin = float(margin)


class WassersteinGanLoss(_ZooLoss):
    """losses.py:55-68."""
    kind = 'wasserstein'


class RaSGANLoss(_ZooLoss):
    """losses.py:87-110 as intended: the reference reads `self.fake_labels` / `self.real_labels`, which do not
    exist (it stores `fake_label` / `real_label`), so its RaSGAN raises on first use; labels 0 / 1."""
    kind = 'rasgan'

    def __init__(self, bce_loss=True):
        if not bce_loss:
            raise NotImplementedError('only the BCE form')


class RaLSGANLoss(_ZooLoss):
    """losses.py:113-133."""
    kind = 'ralsgan'


def _gradient_penalty(discrim, real_x=None, real_xbar=None, fake_x=None, fake_xbar=None, real_cond=None, fake_cond=None,
                      zero_center=False, combine=torch.mean, alpha=None, scale=1.0):
    """losses.py:135-186. alpha ~ U[0,1)^b is drawn on the CPU generator (like the reference) unless
    given. The interpolation, the D forward, the data-gradient sweep (create_graph) and the per-sample
    squared norms all run on the HIP kernels; `input_grads_only` keeps that sweep from computing
    weight gradients nobody asked for."""
    if real_xbar is not None or fake_xbar is not None:
        raise NotImplementedError('sample mappings (TCWYT baseline) are outside the hot path')
    b = real_x.size(0)
    if alpha is None:
        a_dev = TF.draws.alpha(b, real_x.dim(), real_x.device)      # U[0,1)^b from the HOST generator
    else:
        a_dev = alpha.reshape(b).to(device=real_x.device, dtype=torch.float32)
    xh = TF.lerp_rows(a_dev, real_x.detach(), fake_x.detach()).requires_grad_(True)
    ch = None
    if real_cond is not None and fake_cond is not None:
        if real_cond.requires_grad or fake_cond.requires_grad:
            a2 = a_dev.view(b, 1)
            ch = a2 * real_cond + (1 - a2) * fake_cond           # --end2end: keep the text-encoder graph
        else:
            ch = TF.lerp_rows(a_dev, real_cond, fake_cond)
    u, c, _ = discrim(x=xh, cond=ch, xbar=None)
    outs = [u] + ([c] if c is not None else [])
    ones = [TF.ones_like(o) for o in outs]
    with TF.input_grads_only():
        g = torch.autograd.grad(outputs=outs, inputs=[xh], grad_outputs=ones, create_graph=True, retain_graph=True,
                                only_inputs=True)[0]
    sq = TF.row_sqnorm(g)                                        # ||g_b||^2
    if zero_center:
        per = sq
    else:
        per = (torch.sqrt(sq) - 1) ** 2
    if combine is torch.sum:
        return TF.vec_sum(per, scale)
    # a batch MEAN: the average of the ranks' local means already is the global mean — `scale` (world_size under data
    # parallelism) only applies to the sum-combined form above
    return combine(per)


def gradient_penalty(discrim, real_x=None, real_xbar=None, fake_x=None, fake_xbar=None, real_cond=None, fake_cond=None,
                     alphas=None, scale=1.0):
    """losses.py:188-209: multi-scale discriminators use the zero-centred, sum-combined penalty per
    level, summed over levels. `scale` multiplies the result (data-parallel ranks pass world_size so
    that gradient *averaging* reproduces the global-batch SUM)."""
    if hasattr(discrim, 'sub_discrims') and getattr(discrim, 'single_discrim', None) is not None and real_x[0].is_cuda \
            and real_xbar is None:
        # All levels at once through the shared trunk (grouped launches). Same arithmetic as the per-level
        # loop below; the alphas are drawn level by level in the same order.
        n = len(real_x)
        a_dev = [alphas[i].reshape(real_x[i].size(0)).to(real_x[i].device) if alphas is not None else
                 TF.draws.alpha(real_x[i].size(0), real_x[i].dim(), real_x[i].device) for i in range(n)]
        xhs = [TF.lerp_rows(a_dev[i], real_x[i].detach(), fake_x[i].detach()).requires_grad_(True) for i in range(n)]
        chs = None
        if real_cond is not None and fake_cond is not None:
            chs = [TF.lerp_rows(a_dev[i], real_cond[i], fake_cond[i]) for i in range(n)]
        res = discrim(x=xhs, cond=chs, xbar=None)
        outs = []
        for u, c, _ in res:
            outs.append(u)
            if c is not None:
                outs.append(c)
        with TF.input_grads_only():
            gs = torch.autograd.grad(outputs=outs, inputs=xhs, grad_outputs=[TF.ones_like(o) for o in outs],
                                     create_graph=True, retain_graph=True, only_inputs=True)
        return TF.scalar_sum([TF.vec_sum(TF.row_sqnorm(g), scale) for g in gs])
    if hasattr(discrim, 'sub_discrims'):
        gps = []
        for i in range(len(real_x)):
            rc = real_cond[i] if real_cond is not None else None
            fc = fake_cond[i] if real_cond is not None else None
            a = None if alphas is None else alphas[i]
            gps.append(_gradient_penalty(discrim.sub_discrims[i], real_x=real_x[i], fake_x=fake_x[i], real_cond=rc,
                                         fake_cond=fc, zero_center=True, combine=torch.sum, alpha=a, scale=scale))
        return TF.scalar_sum(gps)
    return _gradient_penalty(discrim, real_x=real_x, fake_x=fake_x, real_cond=real_cond, fake_cond=fake_cond, scale=scale)
