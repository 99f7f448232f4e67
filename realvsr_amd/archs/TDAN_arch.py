"""TDAN on the MI355X path: the other in-repo consumer of the modulated DCN
operator and of the fused conv blocks.  Mirrors codes/models/archs/TDAN_arch.py (Align :17-72, Trunk :75-93,
TDAN :96-113): same constructor arguments, forward(x[B, T, C, H, W]) and state_dict keys (the Sequential
holders of the reference are kept so checkpoints load with strict=True); every layer runs through
realvsr_amd.functional.
"""
import math

import torch
import torch.nn as nn

from . import arch_util
from .dcn import ModulatedDeformConvPack as DCN
from .. import functional as RF


class Align(nn.Module):
    def __init__(self, channel=1, nf=64, nb=5, groups=8):
        super(Align, self).__init__()
        self.initial_conv = nn.Conv2d(channel, nf, 3, padding=1, bias=True)
        self.residual_layers = arch_util.make_layer(arch_util.ResidualBlock_noBN, nb)
        self.bottle_neck = nn.Conv2d(nf * 2, nf, 3, padding=1, bias=True)
        self.offset_conv_1 = nn.Conv2d(nf, nf, 3, padding=1, bias=True)
        self.deform_conv_1 = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                                 extra_offset_mask=True)
        self.offset_conv_2 = nn.Conv2d(nf, nf, 3, padding=1, bias=True)
        self.deform_conv_2 = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                                 extra_offset_mask=True)
        self.offset_conv_3 = nn.Conv2d(nf, nf, 3, padding=1, bias=True)
        self.deform_conv_3 = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                                 extra_offset_mask=True)
        self.offset_conv = nn.Conv2d(nf, nf, 3, padding=1, bias=True)
        self.deform_conv = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                               extra_offset_mask=True)
        self.reconstruction = nn.Conv2d(nf, channel, 3, padding=1, bias=True)

    def forward(self, x):
        conv = RF.conv2d
        B, N, C, H, W = x.size()
        out = conv(x.contiguous().view(-1, C, H, W), self.initial_conv, RF.ACT_RELU)
        out = self.residual_layers(out)
        out = out.view(B, N, -1, H, W)
        ref_frame = out[:, N // 2].contiguous()
        y = []
        for i in range(N):
            nei_frame = out[:, i].contiguous()
            fea = conv(ref_frame, self.bottle_neck, x2=nei_frame)   # cat([ref, nei]) is never materialised
            fea = self.deform_conv_1([fea, conv(fea, self.offset_conv_1)])
            fea = self.deform_conv_2([fea, conv(fea, self.offset_conv_2)])
            fea = self.deform_conv_3([nei_frame, conv(fea, self.offset_conv_3)])
            aligned_fea = self.deform_conv([fea, conv(fea, self.offset_conv)])
            y.append(conv(aligned_fea, self.reconstruction))
        return torch.cat(y, dim=1)


class Upsampler(nn.Sequential):
    """codes/models/archs/arch_util.py:142-165 for power-of-two scales without BN / activation: holder of the
    (conv, PixelShuffle) pairs; forward fuses each pair into one kernel."""

    def __init__(self, scale, n_feat, bias=True):
        if scale & (scale - 1):
            raise NotImplementedError('Upsampler: only power-of-two scales are on the MI355X path')
        modules = []
        for _ in range(int(math.log(scale, 2))):
            modules.append(nn.Conv2d(n_feat, 4 * n_feat, 3, padding=1, bias=bias))
            modules.append(nn.PixelShuffle(2))
        super(Upsampler, self).__init__(*modules)

    def forward(self, x):
        for m in self:
            if isinstance(m, nn.Conv2d):
                x = RF.conv2d(x, m, pixel_shuffle=True)
        return x


class Trunk(nn.Module):
    def __init__(self, channel=1, nframes=5, scale=4, nf=64, nb=10):
        super(Trunk, self).__init__()
        self.feature_extractor = nn.Sequential(nn.Conv2d(nframes * channel, 64, 3, padding=1, bias=True),
                                               nn.ReLU(inplace=True))
        self.residual_layers = arch_util.make_layer(arch_util.ResidualBlock_noBN, nb)
        self.upsampler = nn.Sequential(Upsampler(scale, 64), nn.Conv2d(64, 3, 3, padding=1, bias=False))

    def forward(self, x):
        out = RF.conv2d(x, self.feature_extractor[0], RF.ACT_RELU)
        out = self.residual_layers(out)
        out = self.upsampler[0](out)
        return RF.conv2d(out, self.upsampler[1])


class TDAN(nn.Module):
    """Temporally Deformable Alignment Network: (B, T, C, H, W) -> (B, 3, s*H, s*W)."""

    def __init__(self, channel=1, nframes=5, scale=4, nf=64, nb_f=5, nb_b=10, groups=8):
        super(TDAN, self).__init__()
        self.align = Align(channel=channel, nf=nf, nb=nb_f, groups=groups)
        self.trunk = Trunk(channel=channel, nframes=nframes, scale=scale, nf=nf, nb=nb_b)

    def forward(self, x):
        return self.trunk(self.align(x))
