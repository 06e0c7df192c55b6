"""TGANv2 multi-scale generator (unconditional) — same surface / state_dict as
txt2vid/models/tganv2/gen.py:7-119. One process drives one GPU, so the reference's per-block
`data_parallel` scatter/gather disappears (data parallelism is process-level, see txt2vid_amd.dist)."""
import torch
import torch.nn as nn

from .. import layers as L
from ..layers import UpBlock, RenderBlock, Subsample, Linear
from ..conv_lstm import ConvLSTM


class BaseFrameGen(nn.Module):
    """Three UpBlocks 1024 -> 512 -> 256 -> 128 (frames 1x1 -> 8x8); parameter names `up0..up2` as in the reference."""
    WIDTHS = (512, 256)

    def __init__(self, in_channels=1024, out_channels=128):
        super().__init__()
        self.out_channels = 128
        chain = (in_channels,) + self.WIDTHS + (out_channels,)
        for i, (cin, cout) in enumerate(zip(chain[:-1], chain[1:])):
            setattr(self, 'up%d' % i, UpBlock(in_channels=cin, out_channels=cout))

    def forward(self, x, cond=None):
        for i in range(len(self.WIDTHS) + 1):
            x = getattr(self, 'up%d' % i)(x)
        return x


class MultiScaleGen(nn.Module):
    _cond_variant = False

    def __init__(self, latent_size=256, width=128, height=128, num_channels=3, additional_blocks=[64, 32, 32],
                 fm_channels=1024, num_frames=16, cond_dim=0, no_lstm=False):
        super().__init__()
        if no_lstm:
            raise NotImplementedError('no_lstm (TGAN-v1 frame seed generator) is outside the hot path')
        self.no_lstm = False
        self.subsample = Subsample()
        # the recurrent state is a [fm_channels, height/64, width/64] map per sample: 1x1 for 64x64 frames, 2x2 for 128x128
        self.latent_size, self.fm_channels, self.latent_plane_ch = latent_size, fm_channels, fm_channels
        self.fm_width, self.fm_height = max(1, width // 64), max(1, height // 64)
        self.fm_size = self.latent_plane_ch * self.fm_height * self.fm_width
        self.fc = Linear(latent_size + (cond_dim if self._cond_variant else 0), self.fm_size)
        self.clstm = ConvLSTM(input_channels=fm_channels, hidden_channels=[fm_channels], kernel_size=3, step=num_frames,
                              effective_step=range(num_frames))
        # level 0 = BaseFrameGen (-> 128 channels), then one UpBlock per entry of additional_blocks; a RenderBlock per level.
        # The text-conditioned variant puts its 2-D non-local block into the last-but-one UpBlock.
        widths = [128] + list(additional_blocks)
        nl_at = len(additional_blocks) - 1 if self._cond_variant else -1
        levels = [BaseFrameGen()]
        levels += [UpBlock(in_channels=widths[i - 1], out_channels=widths[i], with_non_local=(i == nl_at)) for i in range(1, len(widths))]
        self.abstract_blocks = nn.ModuleList(levels)
        self.render_blocks = nn.ModuleList([RenderBlock(in_channels=w, out_channels=num_channels) for w in widths])
        for weight, taps in self.structurally_live_taps().items():
            weight._t2v_live_taps = list(taps)          # the optimiser only walks these taps of the (tap-major) master copies

    def structurally_live_taps(self):
        """{weight: live kernel taps} for the ConvLSTM when its state is 1x1 (frames below 128x128): a 3x3 kernel on a 1x1
        map only ever multiplies — and only ever receives a gradient through — its centre tap. Used by `dist.model_arena`:
        8 x [1024,1024,3,3] weights = 302 MB of gradients of which 33.5 MB can be non-zero."""
        if self.fm_width != 1 or self.fm_height != 1:
            return {}
        cell = self.clstm.cell0
        k = cell.kernel_size
        centre = (k // 2) * k + k // 2
        return {m.weight: [centre] for m in (cell.Wxi, cell.Wxf, cell.Wxc, cell.Wxo, cell.Whi, cell.Whf, cell.Whc, cell.Who)}

    def forward(self, x, cond=None, return_abstract_maps=False, output_blocks=None):
        from ... import functional as TF
        if cond is not None:
            x = TF.cat_features(x, cond)
        x = self.fc(x)
        x = x.view(x.size(0), self.latent_plane_ch, self.fm_height, self.fm_width)
        hs = self.clstm.forward_stacked(x)                     # [T,B,C,h,w]
        num_frames = hs.size(0)
        x = TF.time_to_batch(hs)                               # [B*T,C,h,w]  (stack+permute+merge_frames)

        abstract, rendered = [], []
        for i in range(len(self.render_blocks)):
            if i != 0 and self.training:
                # [b*T,C,h,w] -> subsample batch and time -> [ceil(b/2)*T/2, C, h, w]
                x, _ = self.subsample_frames(x, num_frames)
                num_frames //= 2
            x = self.abstract_blocks[i](x)
            abstract.append(x)
            if i == len(self.render_blocks) - 1 or self.training or (output_blocks is not None and i in output_blocks):
                if i + 1 < len(self.render_blocks):
                    r, x = self.render_blocks[i](x, fork=True)               # the map also feeds the next level
                else:
                    r = self.render_blocks[i](x)
                rendered.append(TF.frames_to_video(r, num_frames))           # [b,C,T,H,W]
        if return_abstract_maps:
            return rendered, abstract
        return rendered

    def subsample_frames(self, x, num_frames):
        """`Subsample` applied on the merged-frames layout: keeps samples ::2 and frames bt::2
        (gen.py:98-106). bt is drawn from the CPU generator exactly like the reference."""
        from ... import functional as TF
        bt, bt_dev = TF.draws.phase()
        return TF.subsample_frames(x, num_frames, bt, bt_dev), bt
