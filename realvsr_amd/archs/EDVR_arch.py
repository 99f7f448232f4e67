"""EDVR on MI355X: PCD_Align, TSA_Fusion, EDVR, EDVR_NoUp with the reference's constructor
arguments, forward signature and state_dict schema (codes/models/archs/EDVR_arch.py:62-404).

nn.Conv2d modules are kept as parameter holders (names + default init = reference); the compute
is the fused HIP operators of realvsr_amd.functional:
  * every conv + bias + (Leaky)ReLU (+ residual) is one kernel, torch.cat inputs are passed as
    two pointers, ``F.interpolate(...) * 2`` is one kernel, PixelShuffle is folded into the conv;
  * each DCN pack = one conv kernel + one fused DCN kernel (LeakyReLU in its epilogue);
  * TSA's correlation / sigmoid / modulation and its output gate are one kernel each.
The optional ``predeblur`` / ``HR_in`` front ends (EDVR_arch.py:224-231,264-274; enabled by no shipped config) are
built on the same operators (Predeblur_ResNet_Pyramid below) and pinned by tests/golden/edvr_predeblur.npz.
"""
import functools
import os

import torch
import torch.nn as nn

from . import arch_util
from .dcn import ModulatedDeformConvPack as DCN
from .. import functional as RF

LRELU = RF.ACT_LRELU
_USE_SINKS = True   # shared gradient buffers for fan-out tensors (functional.GradSink)


class Predeblur_ResNet_Pyramid(nn.Module):
    """Pre-deblur pyramid (EDVR_arch.py:14-59): 3-level residual pyramid on every frame; same parameter names."""

    def __init__(self, nf=128, HR_in=False):
        super(Predeblur_ResNet_Pyramid, self).__init__()
        self.HR_in = True if HR_in else False
        if self.HR_in:
            self.conv_first_1 = nn.Conv2d(3, nf, 3, 1, 1, bias=True)
            self.conv_first_2 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
            self.conv_first_3 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
        else:
            self.conv_first = nn.Conv2d(3, nf, 3, 1, 1, bias=True)
        basic_block = functools.partial(arch_util.ResidualBlock_noBN, nf=nf)
        for name in ('RB_L1_1', 'RB_L1_2', 'RB_L1_3', 'RB_L1_4', 'RB_L1_5', 'RB_L2_1', 'RB_L2_2', 'RB_L3_1'):
            setattr(self, name, basic_block())
        self.deblur_L2_conv = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
        self.deblur_L3_conv = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)

    def forward(self, x):
        conv, up = RF.conv2d, RF.upsample_bilinear
        if self.HR_in:
            L1_fea = conv(conv(conv(x, self.conv_first_1, LRELU), self.conv_first_2, LRELU), self.conv_first_3, LRELU)
        else:
            L1_fea = conv(x, self.conv_first, LRELU)
        L2_fea = conv(L1_fea, self.deblur_L2_conv, LRELU)
        L3_fea = conv(L2_fea, self.deblur_L3_conv, LRELU)
        L3_fea = up(self.RB_L3_1(L3_fea), 2)
        L2_fea = self.RB_L2_1(L2_fea) + L3_fea
        L2_fea = up(self.RB_L2_2(L2_fea), 2)
        L1_fea = self.RB_L1_2(self.RB_L1_1(L1_fea)) + L2_fea
        return self.RB_L1_5(self.RB_L1_4(self.RB_L1_3(L1_fea)))


class PCD_Align(nn.Module):
    """PCD alignment (EDVR_arch.py:62-132): offsets predicted coarse-to-fine on a 3-level feature pyramid (L3 -> L2 -> L1), one
    modulated DCN per level plus a cascading one; parameter names = the reference's (state_dict schema)."""

    def __init__(self, nf=64, groups=8):
        super(PCD_Align, self).__init__()
        self.L3_offset_conv1 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L3_offset_conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.L3_dcnpack = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                              extra_offset_mask=True)
        self.L2_offset_conv1 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L2_offset_conv2 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L2_offset_conv3 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.L2_dcnpack = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                              extra_offset_mask=True)
        self.L2_fea_conv = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L1_offset_conv1 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L1_offset_conv2 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.L1_offset_conv3 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.L1_dcnpack = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                              extra_offset_mask=True)
        self.L1_fea_conv = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.cas_offset_conv1 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.cas_offset_conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.cas_dcnpack = DCN(nf, nf, 3, stride=1, padding=1, dilation=1, deformable_groups=groups,
                               extra_offset_mask=True)

    def forward(self, nbr_fea_l, ref_fea_l, ref_repeat=None, ref_block=0, sinks=None):
        """nbr_fea_l, ref_fea_l: [L1, L2, L3], each with [B,C,H,W] features.
        Extensions used by EDVR.forward (not in the reference signature):
        ref_repeat = N -- nbr_fea_l holds the N frames of every window frame-major ([N*B, C, H, W]) and ref_fea_l the B
          centre-frame features ONCE (block `ref_block` of nbr_fea_l); the four "concat with the reference" convs then run as
          conv_a(nbr) + conv_b(ref) (functional.conv_cat_bcast) instead of on N repeated copies of ref;
        sinks -- one functional.GradSink per level for nbr_fea_l: its consumers here (offset conv, DCN, reference block)
          accumulate their gradients in one buffer instead of through autograd's adds."""
        conv, up = RF.conv2d, RF.upsample_bilinear
        sk = sinks if (sinks is not None and ref_repeat is not None) else [None, None, None]
        if ref_repeat is not None:
            def cat_ref(x, cv, level, x_sink=None, x_owner=False, premasked=False):
                return RF.conv_cat_bcast(x, ref_fea_l[level], cv, ref_repeat, LRELU, x_sink=x_sink, x_owner=x_owner,
                                         ref_sink=sk[level], ref_block=ref_block, grad_premasked=premasked)
        else:
            def cat_ref(x, cv, level, x_sink=None, x_owner=False, premasked=False):
                return conv(x, cv, LRELU, x2=ref_fea_l[level], grad_premasked=premasked)
        # Single-consumer chains conv -> LeakyReLU -> conv (L3 conv1 -> conv2, L2 / L1 conv2 -> conv3, L1 conv3 -> pack, cascade conv1 -> conv2 -> pack): where the consumer's data-gradient kernel has the mask epilogue (full-resolution
        # level: functional.grad_mask_fusable) the producer's lrelu' is applied THERE (x_premask / feat_premask) and the producer's own
        # backward runs without masks (grad_premasked) -- one read of the activation instead of one in each of its two gradient kernels.
        pm = [(LRELU, 0.1) if RF.grad_mask_fusable(f.shape[2], f.shape[3]) else None for f in nbr_fea_l]
        # level 3 (this conv is the first consumer of the L3 features: it owns their sink)
        L3_offset = cat_ref(nbr_fea_l[2], self.L3_offset_conv1, 2, sk[2], True, premasked=pm[2] is not None)
        L3_offset = conv(L3_offset, self.L3_offset_conv2, LRELU, x_premask=pm[2])   # (its own output has two consumers: the pack and level 2)
        L3_fea = self.L3_dcnpack([nbr_fea_l[2], L3_offset], act=LRELU, sink=sk[2])
        # level 2
        L2_offset = cat_ref(nbr_fea_l[1], self.L2_offset_conv1, 1, sk[1])
        L2_offset = conv(L2_offset, self.L2_offset_conv2, LRELU, x2=up(L3_offset, 2, 2.0), grad_premasked=pm[1] is not None)
        L2_offset = conv(L2_offset, self.L2_offset_conv3, LRELU, x_premask=pm[1])   # (two consumers of its output: the pack and level 1)
        L2_fea = self.L2_dcnpack([nbr_fea_l[1], L2_offset], sink=sk[1])
        L2_fea = conv(L2_fea, self.L2_fea_conv, LRELU, x2=up(L3_fea, 2))
        # level 1
        L1_offset = cat_ref(nbr_fea_l[0], self.L1_offset_conv1, 0, sk[0])
        L1_offset = conv(L1_offset, self.L1_offset_conv2, LRELU, x2=up(L2_offset, 2, 2.0), grad_premasked=pm[0] is not None)
        L1_offset = conv(L1_offset, self.L1_offset_conv3, LRELU, x_premask=pm[0], grad_premasked=pm[0] is not None)
        L1_fea = self.L1_dcnpack([nbr_fea_l[0], L1_offset], sink=sk[0], feat_premask=pm[0])
        L1_fea = conv(L1_fea, self.L1_fea_conv, x2=up(L2_fea, 2))  # no activation (EDVR_arch.py:125)
        # cascading DCN: L1_fea feeds the offset conv (owner of its sink) and the DCN
        cas_sink = RF.GradSink() if (ref_repeat is not None and sinks is not None) else None
        offset = cat_ref(L1_fea, self.cas_offset_conv1, 0, cas_sink, True, premasked=pm[0] is not None)
        offset = conv(offset, self.cas_offset_conv2, LRELU, x_premask=pm[0], grad_premasked=pm[0] is not None)
        return self.cas_dcnpack([L1_fea, offset], act=LRELU, sink=cas_sink, feat_premask=pm[0])


class TSA_Fusion(nn.Module):
    """TSA fusion (EDVR_arch.py:135-208): per-frame correlation with the centre frame gates the aligned features, a 1x1 conv
    fuses them, a 3-level max/avg-pool pyramid produces the spatial gate; parameter names = the reference's."""

    def __init__(self, nf=64, nframes=5, center=2):
        super(TSA_Fusion, self).__init__()
        self.center = center
        self.tAtt_1 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.tAtt_2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.fea_fusion = nn.Conv2d(nframes * nf, nf, 1, 1, bias=True)
        self.sAtt_1 = nn.Conv2d(nframes * nf, nf, 1, 1, bias=True)
        self.sAtt_2 = nn.Conv2d(nf * 2, nf, 1, 1, bias=True)
        self.sAtt_3 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.sAtt_4 = nn.Conv2d(nf, nf, 1, 1, bias=True)
        self.sAtt_5 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.sAtt_L1 = nn.Conv2d(nf, nf, 1, 1, bias=True)
        self.sAtt_L2 = nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=True)
        self.sAtt_L3 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.sAtt_add_1 = nn.Conv2d(nf, nf, 1, 1, bias=True)
        self.sAtt_add_2 = nn.Conv2d(nf, nf, 1, 1, bias=True)

    def forward(self, aligned_fea, center_fea=None, frame_major=False):
        """aligned_fea: [B, N, C, H, W] (reference signature).  Two optional extensions used by EDVR.forward:
        center_fea -- the tensor that was stacked at index `center` (avoids slicing the stack);
        frame_major -- aligned_fea is [N, B, C, H, W], the frame-major batch of the alignment stage as it is: the
        temporal-attention front then runs as one fused autograd node (functional.tsa_temporal_block), no transposing copy."""
        conv, up = RF.conv2d, RF.upsample_bilinear
        if frame_major:
            aligned_fea = RF.tsa_temporal_block(aligned_fea, self.center, self.tAtt_1, self.tAtt_2)  # [B, N*C, H, W]
        else:
            B, N, C, H, W = aligned_fea.size()  # N video frames
            aligned_fea = aligned_fea.contiguous()
            emb_ref = conv(aligned_fea[:, self.center] if center_fea is None else center_fea, self.tAtt_2)
            emb = conv(aligned_fea.view(-1, C, H, W), self.tAtt_1).view(B, N, -1, H, W)
            aligned_fea = RF.tsa_temporal(emb, emb_ref, aligned_fea)  # [B, N*C, H, W]
        # the modulated features feed two 1x1 convs: one gradient buffer for both (fea_fusion, created first, owns it)
        sk = RF.GradSink() if (_USE_SINKS and torch.is_grad_enabled() and aligned_fea.requires_grad) else None
        fea = conv(aligned_fea, self.fea_fusion, LRELU, sink=sk)
        att = conv(aligned_fea, self.sAtt_1, LRELU, dep_sink=sk)
        att = conv(RF.maxavgpool(att), self.sAtt_2, LRELU)
        att_L = conv(att, self.sAtt_L1, LRELU)
        att_L = conv(RF.maxavgpool(att_L), self.sAtt_L2, LRELU)
        att_L = up(conv(att_L, self.sAtt_L3, LRELU), 2)
        att = conv(att, self.sAtt_3, LRELU, residual=att_L)
        att = up(conv(att, self.sAtt_4, LRELU), 2)
        att = conv(att, self.sAtt_5)
        att_add = conv(conv(att, self.sAtt_add_1, LRELU), self.sAtt_add_2)
        return RF.tsa_output(fea, att, att_add)


class _EDVRBase(nn.Module):
    upscale = True

    def __init__(self, nf=64, nc=3, nframes=5, groups=8, front_RBs=5, back_RBs=10, center=None, predeblur=False,
                 HR_in=False, w_TSA=True):
        super(_EDVRBase, self).__init__()
        self.nf = nf
        self.nc = nc
        self.center = nframes // 2 if center is None else center
        # EDVR_NoUp stores these two flags but never acts on them (EDVR_arch.py:330-331, 359-363)
        self.is_predeblur = True if predeblur else False
        self.HR_in = True if HR_in else False
        self.w_TSA = w_TSA
        ResidualBlock_noBN_f = functools.partial(arch_util.ResidualBlock_noBN, nf=nf)
        if self.upscale and self.is_predeblur:
            self.pre_deblur = Predeblur_ResNet_Pyramid(nf=nf, HR_in=self.HR_in)
            self.conv_1x1 = nn.Conv2d(nf, nf, 1, 1, bias=True)
        elif self.upscale and self.HR_in:
            self.conv_first_1 = nn.Conv2d(nc, nf, 3, 1, 1, bias=True)
            self.conv_first_2 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
            self.conv_first_3 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
        else:
            self.conv_first = nn.Conv2d(nc, nf, 3, 1, 1, bias=True)
        self.feature_extraction = arch_util.make_layer(ResidualBlock_noBN_f, front_RBs)
        self.fea_L2_conv1 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
        self.fea_L2_conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.fea_L3_conv1 = nn.Conv2d(nf, nf, 3, 2, 1, bias=True)
        self.fea_L3_conv2 = nn.Conv2d(nf, nf, 3, 1, 1, bias=True)
        self.pcd_align = PCD_Align(nf=nf, groups=groups)
        if self.w_TSA:
            self.tsa_fusion = TSA_Fusion(nf=nf, nframes=nframes, center=self.center)
        else:
            self.tsa_fusion = nn.Conv2d(nframes * nf, nf, 1, 1, bias=True)
        self.recon_trunk = arch_util.make_layer(ResidualBlock_noBN_f, back_RBs)
        if self.upscale:
            self.upconv1 = nn.Conv2d(nf, nf * 4, 3, 1, 1, bias=True)
            self.upconv2 = nn.Conv2d(nf, 64 * 4, 3, 1, 1, bias=True)
            self.pixel_shuffle = nn.PixelShuffle(2)
        self.HRconv = nn.Conv2d(64, 64, 3, 1, 1, bias=True)
        self.conv_last = nn.Conv2d(64, nc, 3, 1, 1, bias=True)

    def extract_features(self, frames, sinks=None):
        """Per-frame part of the network (EDVR_arch.py:275-289): conv_first, the front residual blocks and the
        L2/L3 pyramid convs on a [M, C, H, W] stack of frames.  It does not depend on which window a frame is in,
        which is what the sliding-window driver (realvsr_amd/infer.py) exploits."""
        conv = RF.conv2d
        if self.upscale and self.is_predeblur:      # EDVR_arch.py:264-268
            L1_fea = conv(self.pre_deblur(frames), self.conv_1x1)
        elif self.upscale and self.HR_in:           # :270-274: two stride-2 convs bring HR frames to the LR grid
            L1_fea = conv(conv(conv(frames, self.conv_first_1, LRELU), self.conv_first_2, LRELU), self.conv_first_3, LRELU)
        else:
            L1_fea = conv(frames, self.conv_first, LRELU)
        L1_fea = self.feature_extraction(L1_fea)
        # sinks (training forward only): the stride-2 convs are the FIRST consumers of L1_fea / L2_fea, i.e. the owners of the
        # gradient sinks their later consumers in PCD_Align deposit into (functional.GradSink)
        if sinks is not None:
            sinks[0].shape = tuple(L1_fea.shape)
        L2_fea = conv(L1_fea, self.fea_L2_conv1, LRELU, sink=sinks[0] if sinks is not None else None)
        L2_fea = conv(L2_fea, self.fea_L2_conv2, LRELU)
        if sinks is not None:
            sinks[1].shape = tuple(L2_fea.shape)
        L3_fea = conv(L2_fea, self.fea_L3_conv1, LRELU, sink=sinks[1] if sinks is not None else None)
        L3_fea = conv(L3_fea, self.fea_L3_conv2, LRELU)
        if sinks is not None:
            sinks[2].shape = tuple(L3_fea.shape)
        return L1_fea, L2_fea, L3_fea

    def align_fuse_reconstruct(self, L1_l, L2_l, L3_l, x_center):
        """Window part (EDVR_arch.py:291-320): PCD alignment of every frame to the centre one, TSA fusion,
        reconstruction.  L*_l are lists of N per-frame feature tensors [B, nf, h, w] (contiguous); x_center is
        the centre LR frame [B, C, H, W]."""
        N = len(L1_l)
        B, _, H, W = L1_l[0].shape
        # one PCD call on the frame-major N*B batch (see forward()): fewer launches / tail waves for small frames (7.2 -> 4.8 ms
        # per 180x320 frame) and, at every size, the reference-feature half of the four concat convs runs once per window
        # instead of once per frame (functional.conv_cat_bcast)
        nbr_l = [torch.cat(list(L1_l), 0), torch.cat(list(L2_l), 0), torch.cat(list(L3_l), 0)]
        ref_l = [L1_l[self.center], L2_l[self.center], L3_l[self.center]]
        aligned_nb = self.pcd_align(nbr_l, ref_l, ref_repeat=N).view(N, B, -1, H, W)
        return self._fuse_reconstruct(aligned_nb, x_center)

    def _fuse_reconstruct(self, aligned_nb, x_center):
        """TSA fusion (or the 1x1 fusion conv) + reconstruction on the frame-major aligned features [N, B, C, H, W]."""
        conv = RF.conv2d
        N, B, _, H, W = aligned_nb.shape
        if self.w_TSA:
            fea = self.tsa_fusion(aligned_nb, frame_major=True)
        else:   # what torch.stack(dim=1).view(B, -1, H, W) built (EDVR_arch.py:305-308)
            fea = conv(aligned_nb.transpose(0, 1).reshape(B, -1, H, W), self.tsa_fusion)
        out = self.recon_trunk(fea)
        # single-consumer chains upconv2 -> HRconv -> conv_last: lrelu' of the producer in the consumer's data-gradient epilogue
        # (functional.conv2d x_premask / grad_premasked; the pixel-unshuffle data gradient of upconv2 itself has no such epilogue)
        if self.upscale:
            pm = (LRELU, 0.1) if RF.grad_mask_fusable(4 * H, 4 * W) else None
            out = conv(out, self.upconv1, LRELU, pixel_shuffle=True)
            out = conv(out, self.upconv2, LRELU, pixel_shuffle=True, grad_premasked=pm is not None)
            out = conv(out, self.HRconv, LRELU, x_premask=pm, grad_premasked=pm is not None)
            base = x_center if self.HR_in else RF.upsample_bilinear(x_center, 4)   # EDVR_arch.py:314-317
        else:
            pm = (LRELU, 0.1) if RF.grad_mask_fusable(H, W) else None
            out = conv(out, self.HRconv, LRELU, grad_premasked=pm is not None)
            base = x_center
        return conv(out, self.conv_last, residual=base, x_premask=pm)

    def forward(self, x):
        B, N, C, H, W = x.size()  # N video frames
        hr = self.upscale and self.HR_in
        if H % (16 if hr else 4) or W % (16 if hr else 4):
            raise RuntimeError('EDVR needs H and W divisible by %d (got %dx%d)' % (16 if hr else 4, H, W))
        x_center = x[:, self.center, :, :, :].contiguous()
        # Frame-major batch: frame i of every window is one contiguous block of the [N*B, ...] feature tensors.  The
        # reference loops over the N frames and slices `[:, i]` out of a batch-major view (EDVR_arch.py:291-303);
        # PCD_Align shares its weights across frames, so here all N alignments run as ONE call on the N*B batch
        # (neighbour features = the feature tensors themselves, reference features = the centre block repeated):
        # per-sample arithmetic is unchanged, the kernels are 5x larger (fewer tail waves, 5x fewer launches and
        # weight packs) and autograd sees no per-frame slices (each `[:, i]` costs a full-size zero fill and add in
        # select-backward).
        xf = x.transpose(0, 1).contiguous().view(N * B, C, H, W)
        sinks = [RF.GradSink() for _ in range(3)] if (torch.is_grad_enabled() and _USE_SINKS) else None
        nbr_l = list(self.extract_features(xf, sinks))
        if hr:
            H, W = H // 4, W // 4
        ref_l = [f.view(N, B, *f.shape[1:])[self.center] for f in nbr_l]   # centre-frame features, ONCE (contiguous block)
        aligned = self.pcd_align(nbr_l, ref_l, ref_repeat=N, ref_block=self.center, sinks=sinks)   # [N*B, nf, H, W], frame-major
        return self._fuse_reconstruct(aligned.view(N, B, -1, H, W), x_center)


class EDVR(_EDVRBase):
    """EDVR with x4 upsampling tail (EDVR_arch.py:211-320)."""
    upscale = True


class EDVR_NoUp(_EDVRBase):
    """EDVR without upsampling -- the variant RealVSR trains (EDVR_arch.py:323-404); as in the
    reference it only works for nf=64 (HRconv is hard-coded 64->64, :352)."""
    upscale = False
