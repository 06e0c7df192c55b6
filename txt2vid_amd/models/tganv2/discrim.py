"""Multi-scale discriminator: one Resnet3D shared by the 4 pyramid levels —
txt2vid/models/tganv2/discrim.py:7-31 (state_dict keys `single_discrim.*`)."""
import torch.nn as nn

from ..resnet3d import Resnet3D


class MultiScaleDiscrim(nn.Module):
    _wrap = False

    def __init__(self, discrim_down_blocks=[4, 4, 4, 4], num_channels=3, cond_dim=0, underlying_discrim=Resnet3D,
                 single_discrim=True):
        super().__init__()
        if single_discrim:
            d = underlying_discrim(cond_dim=cond_dim, num_down_blocks=discrim_down_blocks[-1], num_channels=num_channels)
            self.single_discrim = _ModuleWrap(d) if self._wrap else d
            self.sub_discrims = [self.single_discrim for _ in discrim_down_blocks]
        else:
            self.single_discrim = None
            subs = []
            for db in discrim_down_blocks:
                d = underlying_discrim(cond_dim=cond_dim, num_down_blocks=db, num_channels=num_channels)
                subs.append(_ModuleWrap(d) if self._wrap else d)
            self.sub_discrims = nn.ModuleList(subs)

    def forward(self, x=None, cond=None, xbar=None, computed_features=None):
        if self.single_discrim is not None and xbar is None and len(x) > 1 and x[0].is_cuda:
            trunk = self.single_discrim.module if isinstance(self.single_discrim, _ModuleWrap) else self.single_discrim
            if hasattr(trunk, 'groupable') and trunk.groupable():
                return trunk.forward_levels(list(x), None if cond is None else list(cond))
        out = []
        for i, r in enumerate(x):
            c = cond[i] if cond is not None else None
            # `computed_features` is accepted and ignored exactly like the reference (its `cf_i` is never
            # assigned, tganv2_cond/discrim.py:35,40-41), so results are identical.
            out.append(self.sub_discrims[i](r, cond=c, xbar=None if xbar is None else xbar[i]))
        return out


class _ModuleWrap(nn.Module):
    """Stands in for the `nn.DataParallel` wrapper of the conditional discriminator so that the
    checkpoint keys stay `single_discrim.module.*`. No scatter/gather: one process = one GPU."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)
