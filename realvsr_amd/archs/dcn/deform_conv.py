"""Drop-in for codes/models/archs/dcn/deform_conv.py on MI355X.

Same public names, constructor arguments, parameter names (``weight``, ``bias``,
``conv_offset_mask.{weight,bias}``) and initialisation as the reference, so reference
checkpoints load with ``strict=True``.  The arithmetic runs in librealvsr_hip.so
(realvsr_amd/csrc/dcn_kernels.hip) through the C ABI in include/realvsr_hip.h.

* ``modulated_deform_conv`` / ``ModulatedDeformConv``: the dense-offset operator
  (deform_conv.py:97-153, 228-254) -> rvsr_modulated_deform_conv_{forward,backward}.
* ``ModulatedDeformConvPack`` (deform_conv.py:257-292): conv_offset_mask runs as a fused conv
  kernel and the DCN consumes its raw 3*dg*9-channel output directly (chunk / cat / sigmoid are
  addressing + an in-kernel sigmoid) -> rvsr_dcn_pack_{forward,backward}.  ``act`` lets the
  caller fuse the LeakyReLU that follows the pack in PCD_Align (EDVR_arch.py:107,130).
* DCNv1 (``deform_conv`` / ``DeformConv`` / ``DeformConvPack``, deform_conv.py:15-95,156-226): imported by no
  architecture in the reference; the reference's v1 kernels are its modulated kernels without the
  mask factor and without a bias (kernel.cu:190-465 vs :571-767), so the operator runs the same HIP kernels on a mask of
  ones -> rvsr_deform_conv_{forward,backward_input,backward_parameters}.
* Every other geometry (kernel sizes other than 3 x 3, anisotropic stride / padding / dilation, groups, channels per deformable group that
  neither divide nor are a multiple of 8) and the element types f64 / f16 run on the operator's general path (csrc/dcn_generic.hip,
  rvsr_deform_conv_generic_{forward,backward}): columns + GEMM per batch element, as the reference computes every call.
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from ... import functional as RF
from ...functional import DeformConvFunction, ModulatedDeformConvFunction, deform_conv, modulated_deform_conv


def _offset_conv(x, conv):
    """The pack's offset convolution.  3 x 3 / padding 1 / isotropic stride: the fused conv kernel.  Any other geometry (the conv has the
    DCN's own kernel size, stride and padding, deform_conv.py:212-217, 262-268): the general deformable path with ZERO offsets, which is that
    convolution exactly (every sample falls on a pixel centre)."""
    k, st, pd, dl = _pair(conv.kernel_size), _pair(conv.stride), _pair(conv.padding), _pair(conv.dilation)
    if k == (3, 3) and pd == (1, 1) and dl == (1, 1) and st[0] == st[1] and st[0] in (1, 2) and conv.groups == 1 and x.dtype == torch.float32:
        return RF.conv2d(x, conv)
    Ho = (x.shape[2] + 2 * pd[0] - (dl[0] * (k[0] - 1) + 1)) // st[0] + 1
    Wo = (x.shape[3] + 2 * pd[1] - (dl[1] * (k[1] - 1) + 1)) // st[1] + 1
    zero = x.new_zeros(x.shape[0], 2 * k[0] * k[1], Ho, Wo)
    out = deform_conv(x, zero, conv.weight, st, pd, dl, conv.groups, 1)
    return out if conv.bias is None else out + conv.bias.view(1, -1, 1, 1)


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super(DeformConv, self).__init__()
        assert not bias
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        stdv = 1. / math.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1])
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class DeformConvPack(DeformConv):
    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        # (deform_conv.py:222-226) conv_offset is an ordinary convolution with the DCN's own kernel size / stride / padding
        offset = _offset_conv(x, self.conv_offset)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModulatedDeformConv, self).__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):  # deform_conv.py:242-249
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    def __init__(self, *args, extra_offset_mask=False, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)
        self.extra_offset_mask = extra_offset_mask
        self.conv_offset_mask = nn.Conv2d(self.in_channels,
                                          self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                          kernel_size=self.kernel_size, stride=_pair(self.stride),
                                          padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):  # deform_conv.py:270-272
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x, act=RF.ACT_NONE, slope=0.1, sink=None, feat_premask=None):
        """act / slope / sink / feat_premask are extensions (fused LeakyReLU epilogue; functional.GradSink of the sampled input; (act, slope)
        of the activation that produced the offset features when this pack is their only consumer, functional.conv2d x_premask)."""
        if self.extra_offset_mask:  # x = [input, features]
            x, feat = x[0], x[1]
        else:
            feat = x
        # The fused kernels cover what the functional operator's fused path covers (RF._dcn_fused_ok: f32, 3 x 3, isotropic geometry, one
        # group, channels per deformable group a multiple or a divisor of 8) with the offset conv on the fused conv kernel (padding 1,
        # stride 1 / 2); everything else falls through to the reference's wiring on the operator
        st, pd, dl = RF._pair2(self.stride), RF._pair2(self.padding), RF._pair2(self.dilation)
        fused = (self.kernel_size == (3, 3) and pd == (1, 1) and dl == (1, 1) and st[0] in (1, 2) and feat.dtype == torch.float32 and
                 RF._dcn_fused_ok(x, self.weight, st, pd, dl, self.groups, self.deformable_groups))
        if fused:
            om = RF.conv2d(feat, self.conv_offset_mask, x_premask=feat_premask)
            return RF.dcn_pack(x, om, self.weight, self.bias, st[0], pd[0], dl[0], self.deformable_groups, act, slope, sink)
        if feat_premask is not None:
            # the producer of `feat` ran with grad_premasked=True and relies on THIS conv to apply its activation derivative:
            # only the fused branch does
            raise RuntimeError('ModulatedDeformConvPack: feat_premask needs the fused 3x3 / groups=1 path')
        if (self.kernel_size == (3, 3) and pd == (1, 1) and dl == (1, 1) and st[0] == st[1] and st[0] in (1, 2) and
                feat.dtype == torch.float32 and act == RF.ACT_NONE and sink is None):
            # groups > 1, or channels per deformable group the fused kernels do not tile (deform_conv.py:284-292 as written): the unfused
            # wiring on the composed operator, offset conv on the fused conv kernel
            out = RF.conv2d(feat, self.conv_offset_mask)
            o1, o2, mask = torch.chunk(out, 3, dim=1)
            return modulated_deform_conv(x, torch.cat((o1, o2), dim=1), torch.sigmoid(mask), self.weight, self.bias, self.stride,
                                         self.padding, self.dilation, self.groups, self.deformable_groups)
        # Everything the reference's architectures instantiate (EDVR_arch.py:73-74, TDAN_arch.py:29-41) is 3x3, groups=1, "same" padding and
        # runs on the fused kernels above.  Any other geometry: the reference's wiring (deform_conv.py:284-292) on the operator's general path
        # (csrc/dcn_generic.hip), without the fusion extensions.
        if act != RF.ACT_NONE or sink is not None:
            raise RuntimeError('ModulatedDeformConvPack: act / sink are extensions of the fused 3x3 / groups=1 path')
        out = _offset_conv(feat, self.conv_offset_mask)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        return modulated_deform_conv(x, torch.cat((o1, o2), dim=1), torch.sigmoid(mask), self.weight, self.bias, self.stride,
                                     self.padding, self.dilation, self.groups, self.deformable_groups)
