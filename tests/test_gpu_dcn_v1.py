"""DCNv1 (deform_conv / DeformConv / DeformConvPack; reference deform_conv.py:15-95,156-226 -> deform_conv_cuda.cpp:152-488) on the HIP
path against its oracle restatement (oracle.dcn_oracle.deform_conv), through the three C-ABI entry points of include/realvsr_hip.h
section 1b, in both GEMM modes.  -m gpu"""
import pytest
import torch

from gpu_util import check, dev, gemm_modes

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu

# (B, C, Co, dg, H, W, stride, pad, dil, offset std, im2col_step)
SHAPES = [
    (2, 16, 16, 2, 10, 34, 1, 1, 1, 1.0, 64),
    (4, 64, 64, 8, 24, 40, 1, 1, 1, 0.3, 2),      # im2col_step below the batch
    (1, 64, 64, 8, 24, 40, 1, 1, 1, 4.0, 1),      # large offsets: device-selected halos of the backward
    (2, 8, 24, 1, 9, 33, 2, 1, 1, 1.0, 64),       # stride 2
    (1, 16, 16, 2, 11, 13, 1, 2, 2, 1.0, 64),     # dilation 2
    (1, 128, 72, 8, 12, 40, 1, 1, 1, 2.0, 64),    # 16 channels per deformable group, Co > 64
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: '-'.join(str(v) for v in s))
def test_v1_vs_oracle(shape, gemm_mode):
    from oracle.dcn_oracle import deform_conv as oracle_v1
    from realvsr_amd.archs.dcn import deform_conv
    B, C, Co, dg, H, W, stride, pad, dil, ostd, step = shape
    g = torch.Generator().manual_seed(sum(int(v) for v in shape[:9]))
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, Ho, Wo, generator=g) * ostd
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    gout = torch.randn(B, Co, Ho, Wo, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, off, w)]
    oref = oracle_v1(*ref, stride, pad, dil, 1, dg, step)
    oref.backward(gout)
    d = dev()
    got = [t.to(d).requires_grad_(True) for t in (x, off, w)]
    # (pairs, as DeformConv passes them: deform_conv.py:197-199)
    out = deform_conv(*got, (stride, stride), (pad, pad), (dil, dil), 1, dg, step)
    out.backward(gout.to(d))
    torch.cuda.synchronize()
    check('out', out, oref.detach(), 2e-5 if gemm_mode == 'f32' else 1e-4)
    for name, a, r in zip(('grad_input', 'grad_offset', 'grad_weight'), got, ref):
        check(name, a.grad, r.grad, 1e-4)


def test_v1_modules_and_errors(gemm_mode):
    import torch.nn.functional as F
    from realvsr_amd.archs.dcn import DeformConv, DeformConvPack, deform_conv
    d = dev()
    torch.manual_seed(3)
    m = DeformConv(16, 24, 3, stride=1, padding=1, deformable_groups=2).to(d)
    assert tuple(m.weight.shape) == (24, 16, 3, 3) and not hasattr(m, 'bias')
    x = torch.randn(2, 16, 12, 36, device=d)
    out = m(x, torch.zeros(2, 36, 12, 36, device=d))
    check('DeformConv, zero offsets == conv2d', out, F.conv2d(x.double().cpu(), m.weight.double().cpu(), None, padding=1), 2e-5 if gemm_mode == 'f32' else 1e-4)
    p = DeformConvPack(16, 24, 3, stride=1, padding=1, deformable_groups=2).to(d)   # conv_offset is zero-initialised (deform_conv.py:218-220)
    assert sorted(k for k, _ in p.named_parameters()) == ['conv_offset.bias', 'conv_offset.weight', 'weight']
    xg = x.clone().requires_grad_(True)
    out = p(xg)
    check('DeformConvPack at init == conv2d', out, F.conv2d(x.double().cpu(), p.weight.double().cpu(), None, padding=1), 2e-5 if gemm_mode == 'f32' else 1e-4)
    out.sum().backward()
    assert xg.grad is not None and p.weight.grad is not None and p.conv_offset.weight.grad is not None
    with pytest.raises(ValueError):
        deform_conv(torch.randn(8, 4, 4, device=d), None, None)
    with pytest.raises(NotImplementedError):
        deform_conv(torch.randn(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.randn(8, 8, 3, 3))
    with pytest.raises(AssertionError):   # im2col_step must divide the batch (deform_conv.py:41)
        deform_conv(torch.randn(3, 8, 8, 8, device=d), torch.zeros(3, 18, 8, 8, device=d), torch.randn(8, 8, 3, 3, device=d), 1, 1, 1, 1, 1, 2)
    # a 5 x 5 kernel: the operator's general path (tests/test_gpu_dcn_generic.py); zero offsets make it the plain convolution
    x5, w5 = torch.randn(1, 8, 8, 8, device=d), torch.randn(8, 8, 5, 5, device=d)
    check('5x5 DCNv1, zero offsets == conv2d', deform_conv(x5, torch.zeros(1, 50, 8, 8, device=d), w5, 1, 2, 1, 1, 1),
          F.conv2d(x5.double().cpu(), w5.double().cpu(), None, padding=2), 2e-5)
    with pytest.raises(RuntimeError):     # an offset tensor of the wrong shape: loud, not silent
        deform_conv(x5, torch.zeros(1, 18, 8, 8, device=d), w5, 1, 2, 1, 1, 1)
