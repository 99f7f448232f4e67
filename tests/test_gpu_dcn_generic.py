"""The deformable operator's general path (csrc/dcn_generic.hip, include/realvsr_hip.h section 1c) against the CPU oracle: kernel sizes
other than 3 x 3, anisotropic stride / padding / dilation, groups, odd channels per deformable group, DCNv1 and DCNv2, f32 / f64 / f16 --
the argument space of the reference operator (deform_conv_cuda.cpp:490-685, 152-488; kernel.cu:781) outside what its architectures use."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, l2_err

pytestmark = pytest.mark.gpu

# (B, C, Co, H, W, kh, kw, stride, padding, dilation, groups, deformable_groups)
CASES = [
    (2, 4, 6, 9, 11, 2, 3, (2, 1), (1, 0), (1, 2), 2, 2),      # anisotropic everything, groups
    (1, 6, 4, 10, 9, 5, 5, 1, 2, 1, 1, 3),                    # 5 x 5, 2 channels per deformable group
    (2, 10, 5, 8, 8, 1, 1, 1, 0, 1, 5, 2),                    # 1 x 1, five groups, 5 channels per deformable group
    (1, 12, 8, 12, 10, 3, 3, (1, 2), (1, 1), (1, 1), 1, 4),   # 3 x 3 with an anisotropic stride only
    (2, 9, 3, 7, 13, 3, 1, 1, (1, 0), 1, 3, 1),               # 3 x 1, cpg 9 (neither a multiple nor a divisor of 8)
]


def _case(case, dtype, seed=0, v1=False):
    B, C, Co, H, W, kh, kw, stride, pad, dil, groups, dg = case
    from torch.nn.modules.utils import _pair
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(pad), _pair(dil)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    off = torch.randn(B, dg * 2 * kh * kw, Ho, Wo, generator=g, dtype=torch.float64) * 1.5
    m = torch.rand(B, dg * kh * kw, Ho, Wo, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C // groups, kh, kw, generator=g, dtype=torch.float64) / (C * kh * kw / groups) ** 0.5
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    gout = torch.randn(B, Co, Ho, Wo, generator=g, dtype=torch.float64)
    ts = [t.to(dtype) for t in (x, off, m, w, b, gout)]
    return ts, (stride, pad, dil, groups, dg)


def _run(fn_mod, fn_v1, ts, geo, device, v1):
    x, off, m, w, b, gout = [t.to(device) for t in ts]
    stride, pad, dil, groups, dg = geo
    leaves = [t.clone().requires_grad_(True) for t in ((x, off, w) if v1 else (x, off, m, w, b))]
    if v1:
        out = fn_v1(leaves[0], leaves[1], leaves[2], stride, pad, dil, groups, dg)
    else:
        out = fn_mod(*leaves, stride, pad, dil, groups, dg)
    out.backward(gout)
    return [out.detach().cpu()] + [t.grad.cpu() for t in leaves]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('v1', [False, True])
def test_generic_f32_vs_oracle(case, v1):
    from oracle import dcn_oracle as O
    from realvsr_amd import functional as RF
    ts, geo = _case(case, torch.float32, seed=3)
    ref = _run(O.modulated_deform_conv, O.deform_conv, ts, geo, 'cpu', v1)
    got = _run(RF.modulated_deform_conv, RF.deform_conv, ts, geo, dev(), v1)
    names = ('out', 'grad_input', 'grad_offset', 'grad_weight') if v1 else ('out', 'grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias')
    for name, a, r in zip(names, got, ref):
        e = l2_err(a, r)
        print('%-12s l2_err %.2e' % (name, e))
        assert e <= 2e-5, (name, e)


@pytest.mark.parametrize('case', CASES[:3])
def test_generic_f64_vs_oracle(case):
    from oracle import dcn_oracle as O
    from realvsr_amd import functional as RF
    ts, geo = _case(case, torch.float64, seed=4)
    ref = _run(O.modulated_deform_conv, O.deform_conv, ts, geo, 'cpu', False)
    got = _run(RF.modulated_deform_conv, RF.deform_conv, ts, geo, dev(), False)
    for name, a, r in zip(('out', 'grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), got, ref):
        assert a.dtype == torch.float64
        e = l2_err(a, r)
        print('%-12s l2_err %.2e' % (name, e))
        assert e <= 1e-12, (name, e)


@pytest.mark.parametrize('case', CASES[:3])
def test_generic_f16_vs_f64_oracle(case):
    """f16 tensors (arithmetic in f32, results rounded to f16): against the f64 oracle evaluated on the SAME f16-rounded inputs."""
    from oracle import dcn_oracle as O
    from realvsr_amd import functional as RF
    ts, geo = _case(case, torch.float16, seed=5)
    ref = _run(O.modulated_deform_conv, O.deform_conv, [t.double() for t in ts], geo, 'cpu', False)
    got = _run(RF.modulated_deform_conv, RF.deform_conv, ts, geo, dev(), False)
    for name, a, r in zip(('out', 'grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), got, ref):
        assert a.dtype == torch.float16
        e = l2_err(a.double(), r)
        print('%-12s l2_err %.2e' % (name, e))
        assert e <= 3e-3, (name, e)


def test_fused_geometry_in_f64_takes_the_general_path_and_agrees_with_the_fused_kernels():
    """A call the fused f32 kernels cover (3 x 3, isotropic, cpg 8), made in f64: the general path; its f32 rounding must agree with the fused
    f32 result to f32 accuracy."""
    from realvsr_amd import functional as RF
    case = (2, 16, 8, 12, 16, 3, 3, 1, 1, 1, 1, 2)
    ts, geo = _case(case, torch.float64, seed=6)
    got64 = _run(RF.modulated_deform_conv, RF.deform_conv, ts, geo, dev(), False)
    got32 = _run(RF.modulated_deform_conv, RF.deform_conv, [t.float() for t in ts], geo, dev(), False)
    for name, a, r in zip(('out', 'grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), got32, got64):
        e = l2_err(a.double(), r)
        print('%-12s l2_err %.2e' % (name, e))
        assert e <= 2e-5, (name, e)


def test_packs_with_other_geometries_vs_reference_wiring():
    """ModulatedDeformConvPack / DeformConvPack with a 5 x 5 kernel and an anisotropic stride: conv_offset(_mask) runs as the general path with
    zero offsets (= that convolution), the rest is the reference's wiring (deform_conv.py:212-226, 274-292) -- against torch's conv2d + the
    CPU oracle."""
    from oracle import dcn_oracle as O
    from realvsr_amd.archs.dcn import DeformConvPack, ModulatedDeformConvPack
    torch.manual_seed(8)
    x = torch.randn(2, 6, 11, 12)
    pack = ModulatedDeformConvPack(6, 4, 5, stride=1, padding=2, dilation=1, groups=1, deformable_groups=3, bias=True)
    with torch.no_grad():
        pack.conv_offset_mask.weight.normal_(0, 0.05)
        pack.conv_offset_mask.bias.normal_(0, 0.3)
    om = F.conv2d(x, pack.conv_offset_mask.weight, pack.conv_offset_mask.bias, stride=1, padding=2)
    o1, o2, mk = torch.chunk(om, 3, dim=1)
    ref = O.modulated_deform_conv(x, torch.cat((o1, o2), 1), torch.sigmoid(mk), pack.weight.detach(), pack.bias.detach(), 1, 2, 1, 1, 3)
    got = pack.to(dev())(x.to(dev()))
    assert l2_err(got, ref) <= 2e-5, l2_err(got, ref)
    p1 = DeformConvPack(6, 4, (3, 2), stride=(2, 1), padding=(1, 0), dilation=1, groups=2, deformable_groups=2, bias=False)
    with torch.no_grad():
        p1.conv_offset.weight.normal_(0, 0.05)
        p1.conv_offset.bias.normal_(0, 0.3)
    off = F.conv2d(x, p1.conv_offset.weight, p1.conv_offset.bias, stride=(2, 1), padding=(1, 0))
    ref1 = O.deform_conv(x, off, p1.weight.detach(), (2, 1), (1, 0), 1, 2, 2)
    got1 = p1.to(dev())(x.to(dev()))
    assert l2_err(got1, ref1) <= 2e-5, l2_err(got1, ref1)


@pytest.mark.parametrize('what', ['tuple_geometry', 'aniso_stride', 'f64', 'cpg12'])
def test_3x3_packs_outside_the_fused_kernels_fall_through(what):
    """a 3 x 3 / groups = 1 pack used to take the fused branch whatever else it was given.  Tuple stride / padding / dilation, an
    anisotropic stride, f64 tensors and 12 channels per deformable group must reach the reference's wiring on the operator (general path)
    -- and isotropic tuples must still reach the fused kernels; forward and input gradient against torch's conv2d + the CPU oracle."""
    from oracle import dcn_oracle as O
    from realvsr_amd.archs.dcn import ModulatedDeformConvPack
    C, Co, dg, stride, dtype, tol = 16, 8, 2, 1, torch.float32, 2e-5
    if what == 'tuple_geometry':
        stride = (1, 1)
    elif what == 'aniso_stride':
        stride = (1, 2)
    elif what == 'f64':
        dtype, tol = torch.float64, 1e-11
    elif what == 'cpg12':
        C, dg = 24, 2
    padding, dilation = ((1, 1), (1, 1)) if isinstance(stride, tuple) else (1, 1)
    torch.manual_seed(12)
    pack = ModulatedDeformConvPack(C, Co, 3, stride=stride, padding=padding, dilation=dilation, groups=1, deformable_groups=dg, bias=True)
    with torch.no_grad():
        pack.conv_offset_mask.weight.normal_(0, 0.05)
        pack.conv_offset_mask.bias.normal_(0, 0.3)
    pack = pack.to(dtype)
    x = torch.randn(2, C, 12, 16, dtype=dtype)
    xr = x.clone().requires_grad_(True)
    om = F.conv2d(xr, pack.conv_offset_mask.weight.detach(), pack.conv_offset_mask.bias.detach(), stride=stride, padding=1)
    o1, o2, mk = torch.chunk(om, 3, dim=1)
    ref = O.modulated_deform_conv(xr, torch.cat((o1, o2), 1), torch.sigmoid(mk), pack.weight.detach(), pack.bias.detach(), stride, padding, dilation, 1, dg)
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5), dtype=dtype)
    ref.backward(gout)
    xd = x.to(dev()).requires_grad_(True)
    got = pack.to(dev())(xd)
    got.backward(gout.to(dev()))
    assert got.dtype == dtype
    e_o, e_g = l2_err(got, ref), l2_err(xd.grad, xr.grad)
    print('%s: out %.2e grad_input %.2e' % (what, e_o, e_g))
    assert e_o <= tol and e_g <= 5 * tol, (what, e_o, e_g)


def test_generic_path_errors_are_loud():
    from realvsr_amd import functional as RF
    d = dev()
    x = torch.randn(1, 4, 6, 6, device=d, dtype=torch.float64)
    w = torch.randn(4, 4, 5, 5, device=d, dtype=torch.float64)
    with pytest.raises(TypeError):      # mixed element types
        RF.modulated_deform_conv(x, torch.zeros(1, 50, 6, 6, device=d), torch.ones(1, 25, 6, 6, device=d, dtype=torch.float64), w, None, 1, 2, 1, 1, 1)
    with pytest.raises(ValueError):     # a kernel larger than the padded input
        RF.modulated_deform_conv(x, torch.zeros(1, 50, 1, 1, device=d, dtype=torch.float64), torch.ones(1, 25, 1, 1, device=d, dtype=torch.float64),
                                 torch.randn(4, 4, 9, 9, device=d, dtype=torch.float64), None, 1, 0, 1, 1, 1)
    with pytest.raises(RuntimeError):   # channels not divisible into the groups
        RF.modulated_deform_conv(x, torch.zeros(1, 50, 6, 6, device=d, dtype=torch.float64), torch.ones(1, 25, 6, 6, device=d, dtype=torch.float64),
                                 torch.randn(3, 2, 5, 5, device=d, dtype=torch.float64), None, 1, 2, 1, 3, 1)
    with pytest.raises(TypeError):      # an element type the operator does not take
        RF.modulated_deform_conv(x.to(torch.bfloat16), torch.zeros(1, 50, 6, 6, device=d, dtype=torch.bfloat16),
                                 torch.ones(1, 25, 6, 6, device=d, dtype=torch.bfloat16), w.to(torch.bfloat16), None, 1, 2, 1, 1, 1)
