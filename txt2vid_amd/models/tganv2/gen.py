"""TGANv2 multi-scale generator (unconditional) — same surface / state_dict as
txt2vid/models/tganv2/gen.py:7-119. One process drives one GPU, so the reference's per-block
`data_parallel` scatter/gather disappears (data parallelism is process-level, see txt2vid_amd.dist)."""
import torch
import torch.nn as nn

from .. import layers as L
from ..layers import UpBlock, RenderBlock, Subsample, Linear
from ..conv_lstm import ConvLSTM


class BaseFrameGen(nn.Module):
    def __init__(self, in_channels=1024, out_channels=128):
        super().__init__()
        self.out_channels = 128
        self.up0 = UpBlock(in_channels=in_channels, out_channels=512)
        self.up1 = UpBlock(in_channels=512, out_channels=256)
        self.up2 = UpBlock(in_channels=256, out_channels=out_channels)

    def forward(self, x, cond=None):
        return self.up2(self.up1(self.up0(x)))


class MultiScaleGen(nn.Module):
    _cond_variant = False

    def __init__(self, latent_size=256, width=128, height=128, num_channels=3, additional_blocks=[64, 32, 32],
                 fm_channels=1024, num_frames=16, cond_dim=0, no_lstm=False):
        super().__init__()
        if no_lstm:
            raise NotImplementedError('no_lstm (TGAN-v1 frame seed generator) is outside the hot path')
        self.subsample = Subsample()
        self.latent_size = latent_size
        self.fm_channels = fm_channels
        self.fm_width = max(1, width // 64)
        self.fm_height = max(1, height // 64)
        self.latent_plane_ch = fm_channels
        self.fm_size = self.fm_width * self.fm_height * self.latent_plane_ch
        self.fc = Linear(latent_size + (cond_dim if self._cond_variant else 0), self.fm_size)
        self.no_lstm = no_lstm
        self.clstm = ConvLSTM(input_channels=self.latent_plane_ch, hidden_channels=[self.fm_channels], kernel_size=3,
                              step=num_frames, effective_step=range(num_frames))
        base = BaseFrameGen()
        self.render_blocks = [RenderBlock(in_channels=base.out_channels, out_channels=num_channels)]
        self.abstract_blocks = [base]
        for i, block in enumerate(additional_blocks):
            prev = self.abstract_blocks[i].out_channels
            nl = self._cond_variant and (i == len(additional_blocks) - 2)
            self.abstract_blocks.append(UpBlock(in_channels=prev, out_channels=block, with_non_local=nl))
            self.render_blocks.append(RenderBlock(in_channels=block, out_channels=num_channels))
        self.abstract_blocks = nn.ModuleList(self.abstract_blocks)
        self.render_blocks = nn.ModuleList(self.render_blocks)

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
