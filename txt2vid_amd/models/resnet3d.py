"""3-D ResNet discriminator trunk with a non-local block and un/conditional heads — same surface and
state_dict as txt2vid/models/resnet3d.py:6-57, every op a gfx950 kernel."""
import torch
import torch.nn as nn

from .. import functional as TF
from .layers import ResidualBlock, DownBlock, Attention3d, Conv3d, AvgPool3d, ReLU, Linear, down_block_levels, nonlocal_levels


class Resnet3D(nn.Module):

    def __init__(self, num_channels=1, mid_ch=64, which_conv=None, which_pool=None, cond_dim=0, num_down_blocks=4,
                 wide=False, with_attn=True):
        super().__init__()
        which_conv = which_conv or Conv3d
        which_pool = which_pool or AvgPool3d
        self.activation = ReLU()
        res_path = nn.Sequential(which_conv(num_channels, mid_ch, 3, 1, padding=1), self.activation,
                                 which_conv(mid_ch, mid_ch, 3, 1, padding=1), which_pool((1, 2, 2), 2))
        skip = nn.Sequential(which_pool((1, 2, 2), 2), which_conv(num_channels, mid_ch, 1))
        self.res_block = ResidualBlock(inner_module=res_path, identity_map=skip)
        down, in_ch, out_ch = [], mid_ch, 128
        for i in range(num_down_blocks):
            down.append(DownBlock(in_channels=in_ch, out_channels=out_ch, which_conv=which_conv, wide=wide))
            if i == 0 and with_attn:
                down.append(Attention3d(out_ch, which_conv=which_conv))
            in_ch, out_ch = out_ch, out_ch * 2
        self.down = nn.ModuleList(down)
        self.fc_uncond = Linear(in_ch, 1)
        if cond_dim > 0:
            self.fc = Linear(in_ch + cond_dim, 1)

    def _stem(self, x):
        """res_block (resnet3d.py:12-19) with the ReLU fused into the second convolution's gather."""
        m = self.res_block.inner_module
        if isinstance(m[0], Conv3d) and isinstance(m[2], Conv3d):
            h = m[0](x)
            if isinstance(m[3], AvgPool3d) and (m[3].kernel_size, m[3].stride, m[3].padding) == ((1, 2, 2), (2, 2, 2), (0, 0, 0)) and \
                    TF.pool_conv_ok([h], m[2].weight, True):
                # conv2 -> pooling as ONE pooled convolution (functional_pool.py), as in `forward_levels`
                z = TF.pool_conv_group([h], m[2].weight, m[2].bias, relu_in=True, stem=True)[0]
                return TF.add(self.res_block.identity_map(x), z)
            h = TF.relu_conv(h, m[2].weight, m[2].bias)
            return TF.add(self.res_block.identity_map(x), m[3](h))
        return self.res_block(x)

    def groupable(self):
        m = self.res_block.inner_module
        return isinstance(m[0], Conv3d) and isinstance(m[2], Conv3d) and all(
            isinstance(d, (DownBlock, Attention3d)) for d in self.down)

    def forward_levels(self, xs, conds=None):
        """All pyramid levels through the shared trunk in lock-step: every convolution layer is ONE grouped
        launch over the levels (same results as level-by-level `forward`, resnet3d.py:38-57)."""
        m = self.res_block.inner_module
        idm = self.res_block.identity_map
        # the clips feed the residual path and the skip path: a grouped fork sums their two gradients (generator step, gradient
        # penalty) in one launch for all levels instead of one autograd-engine add per level
        xs, xs_skip = TF.fork_group(xs)
        hs = TF.conv_group(xs, m[0].weight, m[0].bias)
        pool_h = None
        if isinstance(m[3], AvgPool3d) and isinstance(idm[0], AvgPool3d) and \
                (m[3].kernel_size, m[3].stride, m[3].padding) == ((1, 2, 2), (2, 2, 2), (0, 0, 0)) and TF.pool_conv_ok(hs, m[2].weight, True):
            # conv2 feeds ONLY the pooling: pool(conv3(relu(h))) = stride-2 conv3 of the box-summed activation — 27 taps over the
            # pooled voxels (a quarter of the even-frame MACs), same values up to summation order (functional_pool.py)
            zs = TF.pool_conv_group(hs, m[2].weight, m[2].bias, relu_in=True, stem=True)
            cfg_x = [(idm[0].kernel_size, idm[0].stride, idm[0].padding)] * len(xs)
            ss = TF.conv_group(TF.avg_pool3d_group(xs_skip, cfg_x), idm[1].weight, idm[1].bias)
            hs = TF.add_group(zs, ss)
            pool_h = False
        elif isinstance(m[3], AvgPool3d) and isinstance(idm[0], AvgPool3d):
            pool_h = (m[3].kernel_size, m[3].stride, m[3].padding)
            if pool_h[0][0] == 1 and pool_h[1][0] == 2 and pool_h[2][0] == 0 and TF.even_frames_ok(hs, m[2].weight):
                # AvgPool3d((1,2,2), stride 2) keeps the even frames of conv2 only: compute just those (half the forward GEMM)
                # and pool what is left with a frame stride of 1
                hs = TF.conv_even_frames_group(hs, m[2].weight, m[2].bias, relu_in=True)
                pool_h = (pool_h[0], (1,) + tuple(pool_h[1][1:]), pool_h[2])
            else:
                hs = TF.conv_group(hs, m[2].weight, m[2].bias, relu_in=True)
        else:
            hs = TF.conv_group(hs, m[2].weight, m[2].bias, relu_in=True)
        if pool_h is False:
            pass                                                                  # (pooled convolution above: already summed)
        elif pool_h is not None:
            cfg_h = [pool_h] * len(xs)
            cfg_x = [(idm[0].kernel_size, idm[0].stride, idm[0].padding)] * len(xs)
            ss = TF.conv_group(TF.avg_pool3d_group(xs_skip, cfg_x), idm[1].weight, idm[1].bias)
            hs = TF.avg_pool3d_group(hs, cfg_h, adds=ss)                       # pool + residual add, all levels, one launch
        else:
            ss = TF.conv_group([idm[0](x) for x in xs_skip], idm[1].weight, idm[1].bias)
            hs = [TF.add(s_, m[3](h)) for s_, h in zip(ss, hs)]
        for d in self.down:
            if isinstance(d, DownBlock):
                hs = down_block_levels(d, hs)
            else:
                hs = nonlocal_levels(d, hs)
        feats = TF.sum_spatial_group(hs)                                      # torch.sum(x, [2,3,4]) of every level: one launch
        f_u = f_c = feats
        if conds is not None:
            # three consumers (unconditional head, conditional head, the caller's mismatched-caption head on the returned features):
            # two grouped forks sum their gradients for all levels in two launches instead of two ATen adds per level
            f_u, rest = TF.fork_group(feats)
            f_c, feats = TF.fork_group(rest)
        w5 = self.fc_uncond.weight.view(self.fc_uncond.weight.shape + (1, 1, 1))
        us = TF.conv_group([f.view(f.shape + (1, 1, 1)) for f in f_u], w5, self.fc_uncond.bias)   # all heads: one launch
        cs = self.cond_heads(f_c, conds) if conds is not None else [None] * len(feats)
        return [(u.view(u.shape[0], u.shape[1]), c, feat) for feat, u, c in zip(feats, us, cs)]

    def cond_heads(self, feats, conds):
        """`self.fc(torch.cat((features, cond), 1))` (resnet3d.py:53-55) for several levels / members at once: one launch for
        all concatenations, one for all heads (instead of two copies + one linear per member)."""
        feats, conds = list(feats), list(conds)
        if not (1 <= len(feats) <= TF.MAX_GROUPS) or not feats[0].is_cuda:
            return [self.fc(TF.cat_features(f, c)) for f, c in zip(feats, conds)]
        cats = TF.cat_features_group(feats, conds)
        w5 = self.fc.weight.view(self.fc.weight.shape + (1, 1, 1))
        cs = TF.conv_group([c.view(c.shape + (1, 1, 1)) for c in cats], w5, self.fc.bias)
        return [c.view(c.shape[0], c.shape[1]) for c in cs]

    def forward(self, x=None, cond=None, xbar=None, computed_features=None):
        uncond = None
        if computed_features is None and x is not None and x.is_cuda and self.groupable():
            # one tensor = a group of one: the same launches and the same fused adjoints (forked inputs, conv1 + skip in one
            # Function, grouped non-local block) as the multi-level pass
            return self.forward_levels([x], [cond] if cond is not None else None)[0]
        if computed_features is not None:
            x = computed_features
        else:
            x = self._stem(x)
            for d in self.down:
                x = d(x)
            x = TF.sum_spatial(x)                           # torch.sum(x, [2,3,4])  (resnet3d.py:48)
            computed_features = x
            uncond = self.fc_uncond(x)
        if cond is not None:
            c = self.fc(TF.cat_features(x, cond))
            return uncond, c, computed_features
        return uncond, None, computed_features
