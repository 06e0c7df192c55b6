"""Conditional multi-scale discriminator (second head on cat(feat, cond)) —
txt2vid/models/tganv2_cond/discrim.py:7-48; state_dict keys `single_discrim.module.*`."""
from ..tganv2.discrim import MultiScaleDiscrim as _Base


class MultiScaleDiscrim(_Base):
    _wrap = True
