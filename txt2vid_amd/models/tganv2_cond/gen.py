"""Text-conditioned TGANv2 generator: `cat(z, cond)` into fc, 2-D non-local block after the third
abstract block — txt2vid/models/tganv2_cond/gen.py:22-124."""
from ..tganv2.gen import MultiScaleGen as _Base, BaseFrameGen  # noqa: F401


class MultiScaleGen(_Base):
    _cond_variant = True

    def __init__(self, latent_size=256, width=64, height=64, num_channels=3, additional_blocks=[64, 32, 32],
                 fm_channels=1024, num_frames=16, cond_dim=256, no_lstm=False):
        super().__init__(latent_size=latent_size, width=width, height=height, num_channels=num_channels,
                         additional_blocks=additional_blocks, fm_channels=fm_channels, num_frames=num_frames,
                         cond_dim=cond_dim, no_lstm=no_lstm)
