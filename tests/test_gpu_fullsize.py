"""Size-independent properties of the hot-path kernels at BASELINE config 2's full sizes (40 frames of
64 x 180 x 320 in the per-frame / alignment stage, 8 windows elsewhere), where the CPU oracle would take minutes:
translation equivariance (bit-exact: exercises every tile seam), linearity, batch additivity of the weight
gradient, and the DCN <-> plain-conv identity between two independently written kernels.  -m gpu"""
import pytest
import torch
import torch.nn as nn

from gpu_util import check, dev, gemm_modes

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu
TOL = {'f32': 2e-5, 'bf16x3': 1e-4}


def _conv(cin=64, cout=64, seed=3):
    torch.manual_seed(seed)
    c = nn.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        c.weight.mul_(0.5)
    return c.to(dev())


def test_conv_translation_equivariance_is_bit_exact(gemm_mode):
    """Shifting the input by (3 rows, 5 columns) shifts the output identically away from the borders: every output
    pixel is the same sum in the same order whichever tile / wave / lane computes it."""
    from realvsr_amd import functional as RF
    conv = _conv()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    xs = torch.zeros_like(x)
    xs[:, :, 3:, 5:] = x[:, :, :-3, :-5]
    with torch.no_grad():
        y, ys = RF.conv2d(x, conv, RF.ACT_LRELU), RF.conv2d(xs, conv, RF.ACT_LRELU)
    # ys(r, c) == y(r - 3, c - 5) wherever both 3x3 footprints see the same data (not the last row / column of ys)
    assert torch.equal(ys[:, :, 5:-1, 7:-1], y[:, :, 2:-4, 2:-6])


def test_conv_linearity_and_wgrad_batch_additivity(gemm_mode):
    from realvsr_amd import functional as RF
    conv = _conv()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    b = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    with torch.no_grad():
        ya, yb, yab = RF.conv2d(a, conv), RF.conv2d(b, conv), RF.conv2d(2.0 * a - b, conv)
        bias = conv.bias.view(1, -1, 1, 1)
    check('conv(2a - b) == 2 conv(a) - conv(b)', yab - bias, 2.0 * (ya - bias) - (yb - bias), 5 * TOL[gemm_mode])
    # weight gradient of the whole batch == sum of the two halves
    gout = torch.randn(40, 64, 180, 320, generator=g).to(dev())

    def wgrad(x, go):
        conv.zero_grad()
        RF.conv2d(x, conv, RF.ACT_LRELU).backward(go)
        return conv.weight.grad.clone(), conv.bias.grad.clone()
    gw, gb = wgrad(a, gout)
    gw1, gb1 = wgrad(a[:20].contiguous(), gout[:20].contiguous())
    gw2, gb2 = wgrad(a[20:].contiguous(), gout[20:].contiguous())
    check('gW(batch) == gW(half 1) + gW(half 2)', gw, gw1 + gw2, 5 * TOL[gemm_mode])
    check('gb(batch) == gb(half 1) + gb(half 2)', gb, gb1 + gb2, 5 * TOL[gemm_mode])


def test_dcn_with_zero_offsets_is_the_plain_conv(gemm_mode):
    """Two independently written kernels (fused DCN vs conv block) agree at the alignment-stage size, forward and
    all three gradients that have a plain-conv counterpart."""
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.dcn import modulated_deform_conv
    conv = _conv(seed=5)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 64, 180, 320, generator=g).to(dev())
    gout = torch.randn(8, 64, 180, 320, generator=g).to(dev())
    off = torch.zeros(8, 8 * 18, 180, 320, device=dev(), requires_grad=True)
    msk = torch.ones(8, 8 * 9, 180, 320, device=dev(), requires_grad=True)
    xd = x.clone().requires_grad_(True)
    wd, bd = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    yd = modulated_deform_conv(xd, off, msk, wd, bd, 1, 1, 1, 1, 8)
    yd.backward(gout)
    xc = x.clone().requires_grad_(True)
    conv.zero_grad()
    yc = RF.conv2d(xc, conv)
    yc.backward(gout)
    tol = 5 * TOL[gemm_mode]
    check('forward', yd, yc, tol)
    check('grad_input', xd.grad, xc.grad, tol)
    check('grad_weight', wd.grad, conv.weight.grad, tol)
    check('grad_bias', bd.grad, conv.bias.grad, tol)
    # mask gradient of a unit mask = per-tap contribution; summed over taps and groups it is <gout, y - bias>
    lhs = float(msk.grad.double().sum())
    rhs = float((gout.double() * (yc.detach().double() - conv.bias.detach().view(1, -1, 1, 1).double())).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(rhs) + 1e-3, (lhs, rhs)
