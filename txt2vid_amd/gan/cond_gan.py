"""The GAN operator surface: D / G step losses with matched / mismatched captions — same methods and
kwargs as txt2vid/gan/cond_gan.py:7-217. Scalar combinations of the per-level losses are done by the
`scalar_*` kernels, not by ATen."""
import numpy as np
import os

import torch

from .. import functional as TF
from .losses import gradient_penalty



def _mean_over_levels(loss, fakes, reals):
    """mean over the pyramid levels of loss(fake, real) (cond_gan.py:121-154). A loss object that offers `<name>_levels`
    (RSGANLoss) gets all levels in one launch; same value as the per-level calls + `scalar_mean`."""
    fakes, reals = list(fakes), list(reals)
    owner = getattr(loss, '__self__', None)
    fn = getattr(owner, getattr(loss, '__name__', '') + '_levels', None) if owner is not None else None
    if fn is not None and 1 <= len(fakes) <= TF.MAX_GROUPS:
        return fn(fakes=fakes, reals=reals)
    return TF.scalar_mean([loss(fake=f, real=r) for f, r in zip(fakes, reals)])

class CondGan(object):
    def __init__(self, gen=None, discrims=None, cond_encoder=None, discrim_names=None, sample_mapping=None,
                 discrim_lambdas=None, gp_scale=1.0):
        assert gen is not None
        assert discrims is not None and len(discrims) >= 1
        if discrim_names is None:
            discrim_names = ['discrim-%d' % i for i in range(len(discrims))]
        if sample_mapping is not None:
            raise NotImplementedError('sample mappings (TCWYT baseline, cond_gan.py:23-24) are outside the hot path')
        self.gen, self.discrims = gen, discrims
        self.sample_mapping = None
        self.cond_encoder = cond_encoder
        self.discrim_names = discrim_names
        self.discrim_lambdas = discrim_lambdas
        self.gp_scale = gp_scale          # data-parallel: world_size, else 1

    def _discrim_weighted_sum(self, losses):
        """cond_gan.py:26-31: mean over discriminators, or lambda-weighted sum."""
        if self.discrim_lambdas is None:
            return TF.scalar_mean(losses)
        return TF.scalar_sum(losses, weights=self.discrim_lambdas)

    def discrim_forward(self, name=None, discrim=None, real=None, real_mapping=None, fake=None, fake_mapping=None,
                        real_cond=None, fake_cond=None, loss=None, gp_lambda=-1):
        """cond_gan.py:34-87."""
        if loss is not None and real is not None and fake is not None and _fusable(discrim, real, fake):
            return self._discrim_forward_fused(discrim, real, fake, real_cond, fake_cond, loss, gp_lambda)
        fake_pred = real_pred = l = None
        if real_cond is not None and fake_cond is not None:
            real_cc = discrim(x=real, cond=real_cond, xbar=None)
            real_pred = real_cc
            if loss is not None:
                # D(real, mismatched captions): only the conditional head differs from real_cc, so its
                # trunk features are reused (the reference *means* to — it passes computed_features,
                # cond_gan.py:45-48 — but recomputes them, tganv2_cond/discrim.py:35,40-41; same values).
                real_ic = [discrim.sub_discrims[i](cond=fake_cond[i], computed_features=real_cc[i][2])
                           for i in range(len(real))] if hasattr(discrim, 'sub_discrims') else \
                    discrim(x=real, cond=fake_cond, xbar=None)
                fake_cc = discrim(x=fake, cond=real_cond, xbar=None)
                lu = TF.scalar_mean([loss(fake=f[0], real=r[0]) for f, r in zip(fake_cc, real_cc)])
                l1 = TF.scalar_mean([loss(fake=f[1], real=r[1]) for f, r in zip(fake_cc, real_cc)])
                l2 = TF.scalar_mean([loss(fake=f[1], real=r[1]) for f, r in zip(real_ic, real_cc)])
                # (lu + (l1 + l2)/2) / 2
                l = TF.scalar_sum([lu, l1, l2], weights=[0.5, 0.25, 0.25])
        else:
            if real is not None:
                real_pred = [r[0] for r in discrim(x=real, cond=None, xbar=None)]
            if fake is not None:
                fake_pred = [f[0] for f in discrim(x=fake, cond=None, xbar=None)]
            if loss is not None and fake_pred is not None and real_pred is not None:
                l = TF.scalar_mean([loss(fake=f, real=r) for f, r in zip(fake_pred, real_pred)])
        if l is not None and gp_lambda > 0:
            gp = gradient_penalty(discrim, real_x=real, fake_x=fake, real_cond=real_cond, fake_cond=fake_cond,
                                  scale=self.gp_scale)
            l = TF.scalar_sum([l, gp], weights=[1.0, gp_lambda])
        return l, fake_pred, real_pred

    def _discrim_forward_fused(self, discrim, real, fake, real_cond, fake_cond, loss, gp_lambda):
        """The whole D-step forward as ONE lock-step pass over up to 8 tensors that share the trunk: the real
        and generated clips of each pyramid level as one batch (D has no batch-coupled op) and, when the
        gradient penalty is on, the interpolated clips x_hat of each level. Same arithmetic as the branchy
        path above; the GP alphas are drawn from the host generator in the same stream position."""
        n = len(real)
        cond = real_cond is not None and fake_cond is not None
        a_dev = [TF.draws.alpha(real[i].size(0), real[i].dim(), real[i].device) for i in range(n)] if gp_lambda > 0 else None
        xhs, chs = [], None
        if not any(t.requires_grad for t in list(real) + list(fake)) and all(r.shape == f.shape for r, f in zip(real, fake)):
            # detached clips (the D step): the real||fake batches and the interpolates of all levels in ONE launch
            rf, xh = TF.cat_lerp_group(real, fake, a_dev)
            if gp_lambda > 0:
                xhs = [t.requires_grad_(True) for t in xh]
        else:
            rf = [TF.cat_batch(r, f) for r, f in zip(real, fake)]
            if gp_lambda > 0:
                xhs = [TF.lerp_rows(a_dev[i], real[i].detach(), fake[i].detach()).requires_grad_(True) for i in range(n)]
        conds = None
        if cond and not any(t.requires_grad for t in list(real_cond) + list(fake_cond)):
            # detached sentence codes: the doubled captions of all levels in one launch, their interpolates in another
            conds, _ = TF.cat_lerp_group(real_cond, real_cond)
            if gp_lambda > 0:
                _, chs = TF.cat_lerp_group(real_cond, fake_cond, a_dev)
        elif cond:
            conds = [TF.cat_batch(c, c) for c in real_cond]
            if gp_lambda > 0:
                chs = [TF.lerp_rows(a_dev[i], real_cond[i], fake_cond[i]) for i in range(n)]
        res = discrim(x=rf + xhs, cond=(conds + chs) if (cond and xhs) else conds, xbar=None)
        both, gp_res = res[:n], res[n:]
        b = [r.size(0) for r in real]
        u_r, u_f = TF.split_rows_group([o[0] for o in both], b)
        if cond:
            c_r, c_f = TF.split_rows_group([o[1] for o in both], b)
            # D(real, mismatched captions): second head on the real half's trunk features
            trunk = discrim.sub_discrims
            shared = trunk[0].module if hasattr(trunk[0], 'module') and not hasattr(trunk[0], 'cond_heads') else trunk[0]
            if all(t is trunk[0] for t in trunk) and hasattr(shared, 'cond_heads'):
                # one shared trunk: the mismatched-caption heads of every level in two launches
                c_ic = shared.cond_heads(TF.head_rows_group([both[i][2] for i in range(n)], b), list(fake_cond))
            else:
                c_ic = [trunk[i](cond=fake_cond[i], computed_features=TF.head_rows(both[i][2], b[i]))[1] for i in range(n)]
            lu = _mean_over_levels(loss, u_f, u_r)
            c_r1, c_r2 = TF.fork_group(list(c_r))          # the real logits enter two loss terms: one launch sums their gradients
            l1 = _mean_over_levels(loss, c_f, c_r1)
            l2 = _mean_over_levels(loss, c_ic, c_r2)
            l = TF.scalar_sum([lu, l1, l2], weights=[0.5, 0.25, 0.25])
            real_pred = [(u_r[i], c_r[i], TF.head_rows(both[i][2], b[i])) for i in range(n)]
            fake_pred = None
        else:
            l = _mean_over_levels(loss, u_f, u_r)
            real_pred, fake_pred = list(u_r), list(u_f)
        if gp_lambda > 0:
            outs = []
            for u, c, _ in gp_res:
                outs.append(u)
                if c is not None:
                    outs.append(c)
            with TF.input_grads_only():
                gs = torch.autograd.grad(outputs=outs, inputs=xhs, grad_outputs=[TF.ones_cached(o) for o in outs],
                                         create_graph=True, retain_graph=True, only_inputs=True)
            if _GROUPED_GP and all(g.is_cuda for g in gs) and len(gs) <= 8:
                # sum over levels and samples of ||dD/dx_hat||^2 as ONE grouped dot product (and one grouped scale + add in its
                # adjoint) instead of a squared-norm / sum pair per level and their adjoints: the same sum in another order
                ga, gb = TF.fork_group(list(gs))
                gp = TF.scalar_sum([TF.dot_group(ga, gb)], weights=[self.gp_scale])
            else:
                gp = TF.scalar_sum([TF.vec_sum(TF.row_sqnorm(g), self.gp_scale) for g in gs])
            l = TF.scalar_sum([l, gp], weights=[1.0, gp_lambda])
        return l, fake_pred, real_pred

    def gen_step_fused(self, fake=None, real=None, cond=None, loss=None):
        """`all_discrim_forward(real)` + `gen_step(fake, real_pred)` (trainer.py:247-263) as ONE lock-step pass
        of the frozen, freshly updated D over the generated AND the (detached) real clips: identical values,
        half the launches; the backward only runs through the generated half."""
        self.gen.zero_grad()
        if self.cond_encoder is not None:
            self.cond_encoder.zero_grad()
        if cond is not None:
            TF.draws.perm(cond[0].size(0), cond[0].device)        # the reference's all_discrim_forward draws a permutation here
        losses = []
        n = len(fake)
        for name, discrim in zip(self.discrim_names, self.discrims):
            frozen = [p for p in discrim.parameters() if p.requires_grad]
            for p in frozen:
                p.requires_grad_(False)
            try:
                res = discrim(x=list(fake) + [r.detach() for r in real], cond=None if cond is None else list(cond) + list(cond),
                              xbar=None)
            finally:
                for p in frozen:
                    p.requires_grad_(True)
            fk, rl = res[:n], res[n:]
            if cond is None:
                losses.append(_mean_over_levels(loss, [ff[0] for ff in fk], [rr[0].detach() for rr in rl]))
            else:
                lu = _mean_over_levels(loss, [ff[0] for ff in fk], [rr[0].detach() for rr in rl])
                lc = _mean_over_levels(loss, [ff[1] for ff in fk], [rr[1].detach() for rr in rl])
                losses.append(TF.scalar_sum([lc, lu], weights=[0.5, 0.5]))
        return self._discrim_weighted_sum(losses)

    def can_fuse_gen_step(self, fake, real):
        return all(_fusable(d, real, fake) and 2 * len(fake) <= TF.MAX_GROUPS for d in self.discrims)

    def gen_step(self, fake=None, real_pred=None, cond=None, loss=None):
        """cond_gan.py:90-118. The uncond branch uses `ff[0]` (the intended semantics; the reference
        passes the whole tuple and crashes, SURVEY §8a defect 1). D's parameters are frozen for this
        forward: the reference computes their gradients here and zeroes them before they are ever used
        (cond_gan.py:157-158), so skipping the D weight-gradient kernels leaves every result identical."""
        self.gen.zero_grad()
        if self.cond_encoder is not None:
            self.cond_encoder.zero_grad()
        losses = []
        for r, name, discrim in zip(real_pred, self.discrim_names, self.discrims):
            frozen = [p for p in discrim.parameters() if p.requires_grad]
            for p in frozen:
                p.requires_grad_(False)
            try:
                fake_cc = discrim(x=fake, cond=cond, xbar=None)
            finally:
                for p in frozen:
                    p.requires_grad_(True)
            if cond is None:
                losses.append(TF.scalar_mean([loss(fake=ff[0], real=rr) for ff, rr in zip(fake_cc, r)]))
            else:
                lu = TF.scalar_mean([loss(fake=ff[0], real=rr[0]) for ff, rr in zip(fake_cc, r)])
                lc = TF.scalar_mean([loss(fake=ff[1], real=rr[1]) for ff, rr in zip(fake_cc, r)])
                losses.append(TF.scalar_sum([lc, lu], weights=[0.5, 0.5]))
        return self._discrim_weighted_sum(losses)

    def all_discrim_forward(self, fake=None, real=None, cond=None, loss=None, gp_lambda=-1):
        """cond_gan.py:121-154."""
        losses, real_pred, fake_pred = [], [], []
        for name, discrim in zip(self.discrim_names, self.discrims):
            real_cond, fake_cond = cond, None
            if cond is not None:
                perm = TF.draws.perm(real_cond[0].size(0), real_cond[0].device)      # numpy global RNG, like the reference
                fc0 = TF.GatherRows.apply(real_cond[0], perm, False)
                fake_cond = [TF.head_rows(fc0, r.size(0)) for r in real_cond]
            l, f, r = self.discrim_forward(name=name, discrim=discrim, real=real, real_cond=real_cond, fake=fake,
                                           fake_cond=fake_cond, loss=loss, gp_lambda=gp_lambda)
            losses.append(l)
            fake_pred.append(f)
            real_pred.append(r)
        return losses, fake_pred, real_pred

    def discrim_step(self, real=None, fake=None, cond=None, loss=None, gp_lambda=-1):
        """cond_gan.py:156-164."""
        for d in self.discrims:
            d.zero_grad()
        if self.cond_encoder is not None:
            self.cond_encoder.zero_grad()
        losses, _, _ = self.all_discrim_forward(real=real, fake=fake, cond=cond, loss=loss, gp_lambda=gp_lambda)
        return self._discrim_weighted_sum(losses)

    def count_params(self):
        from ..util.misc import count_params
        n = int(np.sum([count_params(d) for d in self.discrims])) + count_params(self.gen)
        if self.cond_encoder is not None:
            n += count_params(self.cond_encoder)
        return n

    def __call__(self, *args, **kwargs):
        return self.gen(*args, **kwargs)

    @property
    def discrims_params(self):
        return [d.parameters() for d in self.discrims]

    def save_dict(self):
        """cond_gan.py:186-196: {'gen', 'cond'?, <D name>...}."""
        res = {'gen': self.gen.state_dict()}
        if self.cond_encoder is not None:
            res['cond'] = self.cond_encoder.state_dict()
        for name, d in zip(self.discrim_names, self.discrims):
            res[name] = d.state_dict()
        return res

    def load_from_dict(self, to_load):
        """cond_gan.py:198-217. Accepts both `single_discrim.*` and `single_discrim.module.*` key styles."""
        self.gen.load_state_dict(to_load['gen'])
        if 'cond' in to_load:
            assert self.cond_encoder is not None
            self.cond_encoder.load_state_dict(to_load['cond'])
        for name, d in zip(self.discrim_names, self.discrims):
            if name in to_load:
                d.load_state_dict(_match_keys(to_load[name], d.state_dict().keys()))


def _can_batch(real, fake):
    return all(r.is_cuda and r.shape == f.shape for r, f in zip(real, fake))


_GROUPED_GP = os.environ.get('T2V_NO_GROUPED_GP') is None


def _fusable(discrim, real, fake):
    """Multi-scale discriminator with ONE shared, groupable trunk, device tensors, <= 4 levels."""
    trunk = getattr(discrim, 'single_discrim', None)
    if trunk is None or not hasattr(discrim, 'sub_discrims'):
        return False
    trunk = getattr(trunk, 'module', trunk)
    if not (hasattr(trunk, 'groupable') and trunk.groupable()):
        return False
    return 1 < len(real) and 2 * len(real) <= TF.MAX_GROUPS and _can_batch(real, fake)


def _match_keys(sd, want):
    want = list(want)
    if set(sd.keys()) == set(want):
        return sd
    def strip(k):
        return k.replace('single_discrim.module.', 'single_discrim.').replace('sub_discrims.', 'sub_discrims.')
    by = {strip(k): v for k, v in sd.items()}
    out = {}
    for k in want:
        if strip(k) not in by:
            return sd
        out[k] = by[strip(k)]
    return out
