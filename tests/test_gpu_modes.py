"""The reduced-term GEMM modes ('bf16x2', 'bf16': realvsr_amd.set_gemm_mode) -- opt-in speed modes of conv_fwd5 / conv_wgrad2 /
dcn_fwd3; the default stays the f32-grade three-term split.

What they are held to:
  * DEFINITION, bit for bit: a product of operands that are already bf16 values has no lo part, so the three-term kernel run on
    pre-rounded operands must produce the very bits of the reduced-term kernel run on the original ones ('bf16x2': weights pre-rounded;
    'bf16': weights and input pre-rounded) -- this pins WHICH term each mode drops;
  * operator error against f64: ~2^-9 per product (bf16 rounding of one / both operands), far from the 1e-5 of the default;
  * the north star's bound on the network: PSNR within 1e-3 dB of the reference path (here: PSNR-Y of the build's and the oracle's
    output against a synthetic 30 dB target, and the error of the residual branch itself, at BASELINE config 2's full window)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from gpu_util import dev, l2_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['bf16x2', 'bf16'])
def speed_mode(request):
    from realvsr_amd import _lib
    old = _lib.get_gemm_mode()
    _lib.set_gemm_mode(request.param)
    yield request.param
    _lib.set_gemm_mode(old)


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _conv_case(seed=5, B=2, C=64, Co=64, H=40, W=64):
    torch.manual_seed(seed)
    conv = nn.Conv2d(C, Co, 3, 1, 1).to(dev())
    x = torch.randn(B, C, H, W, device=dev())
    return conv, x


def test_conv_forward_is_the_three_term_kernel_on_rounded_operands(speed_mode):
    from realvsr_amd import _lib
    from realvsr_amd import functional as RF
    conv, x = _conv_case()
    with torch.no_grad():
        y = RF.conv2d(x, conv, RF.ACT_LRELU)
        _lib.set_gemm_mode('bf16x3')
        conv_r = nn.Conv2d(64, 64, 3, 1, 1).to(dev())
        conv_r.weight.copy_(_bf16(conv.weight))
        conv_r.bias.copy_(conv.bias)
        y_r = RF.conv2d(_bf16(x) if speed_mode == 'bf16' else x, conv_r, RF.ACT_LRELU)
        _lib.set_gemm_mode(speed_mode)
    assert torch.equal(y, y_r), (speed_mode, (y - y_r).abs().max().item())


def test_conv_block_against_f64(speed_mode):
    from realvsr_amd import functional as RF
    conv, x = _conv_case(seed=6)
    x.requires_grad_(True)
    # (no activation: an output that moves by 2e-3 flips the LeakyReLU mask of ~1e-3 of the elements, and the gradient of the computed
    # forward then differs from the reference's by 3e-2 in L2 -- a property of the kink, not of the products under test)
    y = RF.conv2d(x, conv, RF.ACT_NONE)
    g = torch.randn_like(y)
    y.backward(g)
    xd = x.detach().double().requires_grad_(True)
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd, padding=1)
    yd.backward(g.double())
    tol = {'bf16x2': 4e-3, 'bf16': 6e-3}[speed_mode]
    for name, a, b in (('out', y, yd), ('grad_x', x.grad, xd.grad), ('grad_w', conv.weight.grad, wd.grad), ('grad_b', conv.bias.grad, bd.grad)):
        e = l2_err(a, b.float())
        print('%s %-8s l2_err %.3e (tol %.1e)' % (speed_mode, name, e, tol))
        assert e <= tol, (name, e)
    # and it IS a reduced-term product: an f32-grade result would sit at ~1e-6
    assert l2_err(y, yd.float()) > 2e-4


def test_dcn_pack_forward(speed_mode):
    """dcn_fwd3 in the speed modes: 'bf16x2' = the three-term kernel on bf16-rounded weights, bit for bit; 'bf16' additionally rounds the
    sampled column values (not an input one could pre-round): bounded against the three-term result."""
    from realvsr_amd import _lib
    from realvsr_amd import functional as RF
    torch.manual_seed(11)
    B, C, H, W, dg = 2, 64, 40, 64, 8
    x = torch.randn(B, C, H, W, device=dev())
    om = torch.randn(B, 27 * dg, H, W, device=dev())
    om[:, :18 * dg] *= 1.25
    w = (torch.randn(C, C, 3, 3, device=dev()) / 24).requires_grad_(True)   # (needs_input_grad -> the training path, no probe pass)
    b = torch.randn(C, device=dev())
    y = RF.dcn_pack(x, om, w, b, 1, 1, 1, dg, RF.ACT_LRELU, 0.1).detach()
    _lib.set_gemm_mode('bf16x3')
    wr = _bf16(w.detach()).requires_grad_(True)
    y3r = RF.dcn_pack(x, om, wr, b, 1, 1, 1, dg, RF.ACT_LRELU, 0.1).detach()
    y3 = RF.dcn_pack(x, om, w, b, 1, 1, 1, dg, RF.ACT_LRELU, 0.1).detach()
    _lib.set_gemm_mode(speed_mode)
    if speed_mode == 'bf16x2':
        assert torch.equal(y, y3r), (y - y3r).abs().max().item()
    e = l2_err(y, y3)
    print('%s dcn_pack forward against the three-term kernel: l2_err %.3e' % (speed_mode, e))
    assert 2e-4 < e <= 6e-3, e


def test_dcn_pack_backward(speed_mode):
    """dcn_bwdin5 / dcn_bwdw4 in the speed modes against the three-term kernels on the same forward state (one seeded pack, 1.25 px
    offsets): gradients of x, offsets + mask, weight and bias."""
    from realvsr_amd import _lib
    from realvsr_amd import functional as RF
    torch.manual_seed(12)
    B, C, H, W, dg = 2, 64, 40, 64, 8
    x0 = torch.randn(B, C, H, W, device=dev())
    om0 = torch.randn(B, 27 * dg, H, W, device=dev())
    om0[:, :18 * dg] *= 1.25
    w0 = torch.randn(C, C, 3, 3, device=dev()) / 24
    b0 = torch.randn(C, device=dev())
    g = torch.randn(B, C, H, W, device=dev())

    def grads():
        x, om, w, b = (t.clone().requires_grad_(True) for t in (x0, om0, w0, b0))
        RF.dcn_pack(x, om, w, b, 1, 1, 1, dg, RF.ACT_NONE, 0.1).backward(g)
        return x.grad, om.grad, w.grad, b.grad
    got = grads()
    _lib.set_gemm_mode('bf16x3')
    ref = grads()
    _lib.set_gemm_mode(speed_mode)
    for name, a, r in zip(('grad_x', 'grad_offset_mask', 'grad_w', 'grad_b'), got, ref):
        e = l2_err(a, r)
        print('%s %-16s l2_err %.3e' % (speed_mode, name, e))
        assert e <= 6e-3, (name, e)
    assert l2_err(got[0], ref[0]) > 1e-4   # (it is a reduced-term product)


@pytest.mark.parametrize('mode', ['bf16x2', 'bf16', 'f16fp8'])
def test_config2_window_holds_the_north_star_psnr_bound(mode):
    """BASELINE config 2's window (EDVR-M nf64, 5 x 180 x 320, offsets rescaled to 1 px) in a speed mode against the CPU oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import edvr_oracle as O
    from realvsr_amd import _lib
    from realvsr_amd.archs.EDVR_arch import EDVR
    import bench
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    torch.manual_seed(0)
    N, H, W = 5, 180, 320
    net = EDVR(nf=64, nc=3, nframes=N, groups=8, front_RBs=5, back_RBs=10, w_TSA=True)
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'conv_offset_mask.weight' in name:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.01)
    x = torch.rand(1, N, 3, H, W, generator=torch.Generator().manual_seed(1234))
    net = net.to(dev())
    bench.offset_stats(net, x.to(dev()), 1.0)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        out_o = O.edvr_forward(sd, x, nframes=N, groups=8, front_RBs=5, back_RBs=10, w_TSA=True)
    old = _lib.get_gemm_mode()
    _lib.set_gemm_mode(mode)
    try:
        with torch.no_grad():
            out = net(x.to(dev())).cpu()
    finally:
        _lib.set_gemm_mode(old)
    base = F.interpolate(x[:, N // 2], scale_factor=4, mode='bilinear', align_corners=False)
    res_err = float((out - out_o).double().norm() / (out_o - base).double().norm())
    target = out_o[:, 0] + torch.randn(out_o[:, 0].shape, generator=torch.Generator().manual_seed(4321)) * 10 ** (-30 / 20)

    def psnr(a):
        return float(10 * torch.log10(1.0 / (a.double() - target.double()).pow(2).mean()))
    d_psnr = abs(psnr(out[:, 0]) - psnr(out_o[:, 0]))
    print('%s: residual-branch error %.3e, max |out - oracle| %.3e, |dPSNR-Y| against a 30 dB target %.2e dB'
          % (mode, res_err, (out - out_o).abs().max().item(), d_psnr))
    assert d_psnr <= 1e-3
    assert res_err <= 1e-2
    # a trained network's residual branch is O(0.05) of the [0, 1] range: even then this error moves a 30 dB PSNR by
    # 4.34 * (res_err * 0.05 / 10^-1.5)^2 dB
    assert 4.34 * (res_err * 0.05 / 10 ** -1.5) ** 2 <= 1e-3


def test_training_steps_in_the_speed_modes_track_the_default_mode(speed_mode):
    """Three optimizer steps of VideoSRModel (EDVR nf64: the 64-row kernels the modes act in) from the same initial state: the per-step
    losses of a speed mode follow the three-term mode's to ~1e-6 (Adam's sign-like first updates amplify gradient differences in the
    parameters themselves, so those are only bounded loosely)."""
    from realvsr_amd import _lib
    from realvsr_amd.VideoSR_model import create_model
    net = dict(which_model_G='EDVR', nf=64, nc=3, nframes=3, groups=8, front_RBs=2, back_RBs=2, center=None, predeblur=False, HR_in=False,
               w_TSA=True)
    opt = {'model': 'VideoSR_AllPair_YCbCr_Split', 'dist': False, 'gpu_ids': [0], 'is_train': True, 'scale': 4, 'augment': None,
           'network_G': net, 'path': {'pretrain_model_G': None, 'strict_load': True},
           'train': {'pixel_criterion_y': 'lappyr', 'pixel_weight_y': 1.0, 'pixel_criterion_c': 'gw', 'pixel_weight_c': 0.5,
                     'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-4, 'beta1': 0.9, 'beta2': 0.99}}
    g = torch.Generator().manual_seed(21)
    data = {'LQs': torch.rand(2, 3, 3, 32, 64, generator=g), 'GT': torch.rand(2, 3, 3, 128, 256, generator=g)}

    def run(mode):
        _lib.set_gemm_mode(mode)
        torch.manual_seed(5)
        model = create_model(opt)
        with torch.no_grad():
            for name, p in model.netG.named_parameters():
                if 'conv_offset_mask.weight' in name:
                    p.normal_(0, 0.01, generator=torch.Generator(device=p.device).manual_seed(len(name)))
        from realvsr_amd import functional as RF
        RF.invalidate_weight_cache()
        start = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
        losses = []
        for step in range(1, 4):
            model.feed_data(data)
            model.optimize_parameters(step)
            losses.append(model.get_current_log()['l_pix'])
        upd = torch.cat([(v.detach() - start[k]).flatten() for k, v in model.netG.state_dict().items()])
        return losses, upd
    try:
        l_ref, u_ref = run('bf16x3')
        l_got, u_got = run(speed_mode)
    finally:
        _lib.set_gemm_mode(speed_mode)   # (the fixture restores the caller's mode)
    print(speed_mode, 'losses', l_got, 'three-term', l_ref)
    for a, b in zip(l_got, l_ref):
        assert abs(a - b) <= 1e-4 * abs(b), (l_got, l_ref)   # (measured: 1e-6)
    rel = float((u_got - u_ref).norm() / u_ref.norm())
    print('parameter update after 3 steps: rel l2 difference %.3f' % rel)
    assert rel <= 0.2 and torch.isfinite(u_got).all()   # (measured: 0.09)


def _speed_mode_conv_cases():
    from test_gpu_conv import CASES, _random_cases
    # the 3 x 3 stride-1 layers with more than 32 output channels: the 64-row m-block kernels the modes act in (ragged tiles, concat
    # inputs, residuals, PixelShuffle stores, widths on and off the vector path)
    return [c for c in CASES + _random_cases(36, 20260928) if c[3] == 3 and c[4] == 1 and c[2] > 32]


@pytest.mark.parametrize('case', _speed_mode_conv_cases(), ids=lambda c: '-'.join(str(v) for v in c))
def test_conv_shapes_in_the_speed_modes(case, speed_mode):
    """The shape sweep of tests/test_gpu_conv.py (forward, data / concat / residual / weight / bias gradients against f64) in the speed modes,
    at their tolerance: the reduced-term instantiations share every addressing path with the default kernels."""
    from test_gpu_conv import test_conv_block_forward_backward
    test_conv_block_forward_backward(case, speed_mode)


def _speed_mode_dcn_cases():
    from test_gpu_dcn import SHAPES, _random_shapes
    # more than 32 output channels, stride / dilation 1: dcn_fwd3<MT >= 2>, dcn_bwdin5<NK >= 4>, dcn_bwdw4 -- the kernels the modes act in
    return [c for c in SHAPES + _random_shapes(14, 928) if c[2] > 32 and c[6] == 1 and c[8] == 1]


@pytest.mark.parametrize('shape', _speed_mode_dcn_cases(), ids=lambda s: '-'.join(str(v) for v in s))
def test_dcn_shapes_in_the_speed_modes(shape, speed_mode):
    """The DCN operator's shape sweep (tests/test_gpu_dcn.py: ragged tiles, group counts, every offset regime of the backward's window
    selection) in the speed modes, against the oracle at their tolerance."""
    from test_gpu_dcn import test_random_shapes_vs_oracle
    test_random_shapes_vs_oracle(shape, speed_mode)


# ------------------------------------------------------------------------------------------------------------------------------
# 'f16fp8' (round 5): the forward 3 x 3 / stride-1 convs with more than 32 output channels in the f16 + fp8 product format
# (a1*b1 in f16 + (a1*b2 + a2*b1) in fp8 e4m3, a1 = f16(a), a2 = a - a1), everything else as in the default mode.
@pytest.fixture
def f16fp8_mode():
    from realvsr_amd import _lib
    old = _lib.get_gemm_mode()
    _lib.set_gemm_mode('f16fp8')
    assert _lib.get_gemm_mode() == 'f16fp8'
    yield 'f16fp8'
    _lib.set_gemm_mode(old)
    assert _lib.get_gemm_mode() == old


def test_f16fp8_forward_error_and_scope():
    """Where the format acts the forward error against f64 is ~1.2e-5 (default: ~4.6e-6; the two-term mode: 1.7e-3) -- above the default's, which
    shows that the format ran, and below 2.5e-5; where it does not act (32 output channels, 1x1, stride 2, and every gradient of a conv
    without activation) the results are those of the default mode bit for bit."""
    from realvsr_amd import functional as RF, _lib
    d = dev()
    torch.manual_seed(11)
    old = _lib.get_gemm_mode()
    try:
        for (C, Co, k, stride, H, W, acts) in [(64, 64, 3, 1, 64, 96, True), (128, 128, 3, 1, 40, 64, True), (48, 40, 3, 1, 37, 64, True),
                                               (64, 32, 3, 1, 40, 64, False), (64, 64, 1, 1, 40, 64, False), (64, 64, 3, 2, 40, 64, False)]:
            conv = nn.Conv2d(C, Co, k, stride, k // 2).to(d)
            x = torch.randn(2, C, H, W, device=d, requires_grad=True)
            ref = F.conv2d(x.detach().double(), conv.weight.double(), conv.bias.double(), stride=stride, padding=k // 2)
            got = {}
            for mode in ('bf16x3', 'f16fp8'):
                _lib.set_gemm_mode(mode)
                x.grad = None
                conv.weight.grad = None
                y = RF.conv2d(x, conv, RF.ACT_NONE)
                y.backward(torch.ones_like(y))
                got[mode] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone())
            e3, e8 = l2_err(got['bf16x3'][0], ref), l2_err(got['f16fp8'][0], ref)
            print((C, Co, k, stride, H, W), 'forward l2 error against f64: default %.2e, f16fp8 %.2e' % (e3, e8))
            if acts:
                assert 7e-6 < e8 < 2.5e-5 and e3 < 7e-6, (e3, e8)
            else:
                assert torch.equal(got['bf16x3'][0], got['f16fp8'][0])
            assert torch.equal(got['bf16x3'][1], got['f16fp8'][1]) and torch.equal(got['bf16x3'][2], got['f16fp8'][2])   # gradients: the default kernels
    finally:
        _lib.set_gemm_mode(old)


@pytest.mark.parametrize('case', _speed_mode_conv_cases(), ids=lambda c: '-'.join(str(v) for v in c))
def test_conv_shapes_in_the_f16fp8_mode(case, f16fp8_mode):
    """The conv shape sweep (ragged tiles, concat inputs, residuals, PixelShuffle stores, widths on and off the vector path) in the 'f16fp8' mode."""
    from test_gpu_conv import test_conv_block_forward_backward
    test_conv_block_forward_backward(case, 'f16fp8')
