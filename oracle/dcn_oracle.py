"""oracle/dcn_oracle.py -- TEST INFRASTRUCTURE ONLY (not product code).

CPU restatement of the reference's modulated deformable convolution operator:

* device kernels  -> oracle/dcn_oracle.c (im2col / col2im / col2im_coord)
* host code       -> this file, following
    codes/models/archs/dcn/src/deform_conv_cuda.cpp:490-569 (forward: per-sample im2col +
      addmm with weight.flatten(1), bias added last) and
    codes/models/archs/dcn/src/deform_conv_cuda.cpp:571-685 (backward: col_grad = W^T gOut,
      col2im_coord, col2im, im2col recompute, gW += gOut col^T, gBias += gOut 1),
* autograd glue   -> codes/models/archs/dcn/deform_conv.py:97-153 (ModulatedDeformConvFunction).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (realvsr_amd) never does: it raises when its HIP library is missing.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle_dcn.so with gcc (idempotent)."""
    so = os.path.join(_HERE, 'liboracle_dcn.so')
    src = os.path.join(_HERE, 'dcn_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _suffix(t):
    if t.dtype == torch.float32:
        return '_f32'
    if t.dtype == torch.float64:
        return '_f64'
    raise TypeError('oracle DCN supports float32/float64, got %s' % t.dtype)


def _hw(v):
    """(h, w) of a geometry argument given as an int or a pair (the C restatement takes every component separately)."""
    from torch.nn.modules.utils import _pair
    a, b = _pair(v)
    return int(a), int(b)


def _geom(x, weight, stride, padding, dilation):
    kh, kw = weight.shape[2:4]
    H, W = x.shape[2:4]
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(padding), _hw(dilation)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return kh, kw, H, W, Ho, Wo


def _ints(*a):
    return [ctypes.c_int(int(v)) for v in a]


def im2col(x_b, off_b, msk_b, kh, kw, stride, padding, dilation, dg, Ho, Wo):
    C, H, W = x_b.shape
    col = torch.empty(C * kh * kw, Ho * Wo, dtype=x_b.dtype)
    fn = getattr(_lib(), 'oracle_modulated_im2col' + _suffix(x_b))
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(padding), _hw(dilation)
    fn(_ptr(x_b), _ptr(off_b), _ptr(msk_b),
       *_ints(C, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, dg),
       _ptr(col))
    return col


class ModulatedDeformConvOracle(torch.autograd.Function):
    """Same call signature as the reference's ``modulated_deform_conv``
    (deform_conv.py:99-100): (input, offset, mask, weight, bias, stride, padding, dilation,
    groups, deformable_groups)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1,
                groups=1, deformable_groups=1):
        assert not input.is_cuda, 'oracle runs on CPU tensors only'
        x = input.contiguous()
        offset = offset.contiguous()
        mask = mask.contiguous()
        weight = weight.contiguous()
        B, C = x.shape[:2]
        Co = weight.shape[0]
        kh, kw, H, W, Ho, Wo = _geom(x, weight, stride, padding, dilation)
        assert C == weight.shape[1] * groups
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups, bias is not None)
        ctx.save_for_backward(x, offset, mask, weight)
        out = torch.zeros(B, Co, Ho, Wo, dtype=x.dtype)
        cg, og = C // groups, Co // groups
        for b in range(B):  # cpp:539-561 -- one im2col + `groups` GEMMs per sample
            col = im2col(x[b], offset[b], mask[b], kh, kw, stride, padding, dilation,
                         deformable_groups, Ho, Wo)
            for g in range(groups):
                wg = weight[g * og:(g + 1) * og].flatten(1)
                out[b, g * og:(g + 1) * og] = (wg @ col[g * cg * kh * kw:(g + 1) * cg * kh * kw]
                                                ).view(og, Ho, Wo)
        if bias is not None:  # cpp:566-568
            out += bias.view(1, -1, 1, 1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        x, offset, mask, weight = ctx.saved_tensors
        stride, padding, dilation, groups, dg, with_bias = ctx.cfg
        gout = grad_output.contiguous()
        B, C = x.shape[:2]
        Co = weight.shape[0]
        kh, kw, H, W, Ho, Wo = _geom(x, weight, stride, padding, dilation)
        K = kh * kw
        sfx = _suffix(x)
        lib = _lib()
        gx = torch.zeros_like(x)
        goff = torch.zeros_like(offset)
        gmask = torch.zeros_like(mask)
        gw = torch.zeros_like(weight)
        gb = torch.zeros(Co, dtype=x.dtype)
        cg, og = C // groups, Co // groups
        (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(padding), _hw(dilation)
        geo = _ints(C, H, W, Ho, Wo, kh, kw, ph, pw, sh, sw, dh, dw, dg)
        for b in range(B):
            col_grad = torch.empty(C * K, Ho * Wo, dtype=x.dtype)
            for g in range(groups):  # cpp:623-626
                wg = weight[g * og:(g + 1) * og].flatten(1)
                col_grad[g * cg * K:(g + 1) * cg * K] = wg.t() @ gout[b, g * og:(g + 1) * og].flatten(1)
            getattr(lib, 'oracle_modulated_col2im_coord' + sfx)(  # cpp:634-638
                _ptr(col_grad), _ptr(x[b]), _ptr(offset[b]), _ptr(mask[b]), *geo,
                _ptr(goff[b]), _ptr(gmask[b]))
            getattr(lib, 'oracle_modulated_col2im' + sfx)(  # cpp:640-643
                _ptr(col_grad), _ptr(offset[b]), _ptr(mask[b]), *geo, _ptr(gx[b]))
            col = im2col(x[b], offset[b], mask[b], kh, kw, stride, padding, dilation, dg, Ho, Wo)
            for g in range(groups):  # cpp:659-671
                go = gout[b, g * og:(g + 1) * og].flatten(1)
                gw[g * og:(g + 1) * og] += (go @ col[g * cg * K:(g + 1) * cg * K].t()).view(og, cg, kh, kw)
                gb[g * og:(g + 1) * og] += go.sum(1)
        return (gx, goff, gmask, gw, gb if with_bias else None, None, None, None, None, None)


modulated_deform_conv = ModulatedDeformConvOracle.apply


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    """DCNv1 (deform_conv.py:15-95 -> deform_conv_cuda.cpp:152-488 -> kernel.cu:190-465), TEST INFRASTRUCTURE ONLY.

    The reference's v1 device code is its modulated code with the mask factor deleted, line for line:
      deformable_im2col_gpu_kernel        kernel.cu:190-241  ==  modulated_deformable_im2col_gpu_kernel        :571-633 without `* mask` (:621)
      deformable_col2im_gpu_kernel        kernel.cu:279-335  ==  modulated_deformable_col2im_gpu_kernel        :636-693 without `* mask` (:666)
      deformable_col2im_coord_gpu_kernel  kernel.cu:373-430  ==  modulated_deformable_col2im_coord_gpu_kernel  :696-767 without `* mask` (:751) and
                                                                 without the grad_mask output
      helpers :84-188 == :467-568 (same zero-outside bilinear rule, same corner / coordinate weights, same -2 sentinel, same 5x5 window)
    and its host functions run the same per-sample GEMMs (cpp:209-240, 317-352, 436-470 vs :539-561, 617-671) without a bias; `im2col_step` only
    batches the column buffer.  The restatement is therefore the modulated oracle on a mask of ones, with the v1 wrapper's own checks
    (4-D input, im2col_step divides the batch: deform_conv.py:19-21,40-41).  Parity pin: none beyond the modulated oracle's (the reference has no
    CPU path and no test for this operator) -- identities + finite differences in tests/test_oracle_dcn.py."""
    from torch.nn.modules.utils import _pair
    if input is not None and input.dim() != 4:
        raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    cur = min(im2col_step, input.shape[0])
    assert input.shape[0] % cur == 0, 'im2col step must divide batchsize'
    kh, kw = weight.shape[2:]
    Ho = (input.shape[2] + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (input.shape[3] + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    mask = torch.ones(input.shape[0], deformable_groups * kh * kw, Ho, Wo, dtype=input.dtype)
    return ModulatedDeformConvOracle.apply(input, offset, mask, weight, None, (sh, sw), (ph, pw), (dh, dw), groups, deformable_groups)
