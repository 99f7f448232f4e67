"""Pins the CPU oracle of the DCN operator (oracle/dcn_oracle.{c,py}); CPU only.

The reference has no tests/golden vectors for this operator, so the
oracle is pinned by known-answer identities and float64 finite differences, plus determinism
against the committed fixture generated through the imported reference Pack wiring."""
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err
from oracle.dcn_oracle import modulated_deform_conv


def _rand(B=2, C=8, Co=6, dg=2, H=6, W=7, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g, dtype=dtype)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=dtype)
    b = torch.randn(Co, generator=g, dtype=dtype)
    return x, w, b, dg


def test_zero_offset_unit_mask_is_conv2d():
    x, w, b, dg = _rand()
    off = torch.zeros(2, dg * 18, 6, 7)
    m = torch.ones(2, dg * 9, 6, 7)
    out = modulated_deform_conv(x, off, m, w, b, 1, 1, 1, 1, dg)
    assert rel_err(out, F.conv2d(x, w, b, padding=1)) < 1e-6


def test_half_mask_scales_conv():  # zero-initialised conv_offset_mask => sigmoid(0) = 0.5
    x, w, b, dg = _rand(seed=1)
    off = torch.zeros(2, dg * 18, 6, 7)
    m = torch.full((2, dg * 9, 6, 7), 0.5)
    out = modulated_deform_conv(x, off, m, w, b, 1, 1, 1, 1, dg)
    assert rel_err(out, 0.5 * F.conv2d(x, w, None, padding=1) + b.view(1, -1, 1, 1)) < 1e-6


def test_integer_offsets_shift_the_input():
    x, w, b, dg = _rand(seed=2)
    off = torch.zeros(2, dg * 18, 6, 7)
    off[:, 0::2] = 1.0   # dy = +1
    off[:, 1::2] = -2.0  # dx = -2
    m = torch.ones(2, dg * 9, 6, 7)
    out = modulated_deform_conv(x, off, m, w, b, 1, 1, 1, 1, dg)
    shifted = torch.zeros_like(x)  # shifted[h, w] = x[h+1, w-2], zero outside
    shifted[:, :, :-1, 2:] = x[:, :, 1:, :-2]
    # rows/cols whose zero-PADDING taps map to valid shifted samples differ by construction
    assert rel_err(out[:, :, 1:, :-1], F.conv2d(shifted, w, b, padding=1)[:, :, 1:, :-1]) < 1e-6


def test_zero_mask_and_far_offsets_give_bias():
    x, w, b, dg = _rand(seed=3)
    off = torch.zeros(2, dg * 18, 6, 7)
    out = modulated_deform_conv(x, off, torch.zeros(2, dg * 9, 6, 7), w, b, 1, 1, 1, 1, dg)
    assert torch.equal(out, b.view(1, -1, 1, 1).expand_as(out).contiguous())
    out = modulated_deform_conv(x, off + 100.0, torch.ones(2, dg * 9, 6, 7), w, b, 1, 1, 1, 1, dg)
    assert torch.equal(out, b.view(1, -1, 1, 1).expand_as(out).contiguous())


def test_stride_dilation_match_conv2d():
    x, w, b, dg = _rand(H=9, W=11, seed=4)
    for stride, pad, dil in [(2, 1, 1), (1, 2, 2), (2, 0, 1)]:
        ref = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
        Ho, Wo = ref.shape[2:]
        out = modulated_deform_conv(x, torch.zeros(2, dg * 18, Ho, Wo), torch.ones(2, dg * 9, Ho, Wo),
                                    w, b, stride, pad, dil, 1, dg)
        assert rel_err(out, ref) < 1e-6


def test_f64_finite_differences():
    g = torch.Generator().manual_seed(5)
    B, C, Co, dg, H, W = 1, 4, 3, 2, 4, 5
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    # keep sample positions away from the integer lattice (the operator is only piecewise smooth)
    off = (torch.rand(B, dg * 18, H, W, generator=g, dtype=torch.float64) * 0.6 + 0.2)
    off = off * torch.where(torch.rand(off.shape, generator=g) > 0.5, 1.0, -1.0) + \
        torch.randint(-2, 3, off.shape, generator=g).double()
    off.requires_grad_(True)
    m = torch.rand(B, dg * 9, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(Co, generator=g, dtype=torch.float64, requires_grad=True)
    fn = lambda *a: modulated_deform_conv(*a, 1, 1, 1, 1, dg)  # noqa: E731
    assert torch.autograd.gradcheck(fn, (x, off, m, w, b), eps=1e-6, atol=1e-6, rtol=1e-5)


def test_matches_committed_fixture():
    g = load_golden('dcn_op')
    t = {k: torch.from_numpy(v) for k, v in g.items() if v.ndim > 0}
    leaves = [t[k].clone().requires_grad_(True) for k in ('x', 'offset', 'mask', 'weight', 'bias')]
    out = modulated_deform_conv(*leaves, 1, 1, 1, 1, int(g['dg']))
    out.backward(t['gout'])
    assert rel_err(out, t['out']) < 1e-6
    for leaf, key in zip(leaves, ('gx', 'goffset', 'gmask', 'gweight', 'gbias')):
        assert rel_err(leaf.grad, t[key]) < 2e-6, key


# ---- DCNv1 restatement (oracle.dcn_oracle.deform_conv): the modulated oracle on a mask of ones + the v1 wrapper's checks
def test_v1_zero_offsets_is_conv2d_and_checks():
    import pytest
    from oracle.dcn_oracle import deform_conv
    x, w, _, dg = _rand(seed=7)
    out = deform_conv(x, torch.zeros(2, dg * 18, 6, 7), w, 1, 1, 1, 1, dg)
    assert rel_err(out, F.conv2d(x, w, None, padding=1)) < 1e-6
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    out = deform_conv(x, torch.zeros(2, dg * 18, *ref.shape[2:]), w, (2, 2), (1, 1), (1, 1), 1, dg, 1)
    assert rel_err(out, ref) < 1e-6
    with pytest.raises(ValueError):
        deform_conv(x[0], torch.zeros(dg * 18, 6, 7), w)
    with pytest.raises(AssertionError):
        deform_conv(torch.cat([x, x[:1]]), torch.zeros(3, dg * 18, 6, 7), w, 1, 1, 1, 1, dg, 2)   # im2col_step 2 does not divide 3


def test_v1_f64_finite_differences():
    from oracle.dcn_oracle import deform_conv
    g = torch.Generator().manual_seed(8)
    B, C, Co, dg, H, W = 1, 4, 3, 2, 4, 5
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    off = (torch.rand(B, dg * 18, H, W, generator=g, dtype=torch.float64) * 0.6 + 0.2)
    off = off * torch.where(torch.rand(off.shape, generator=g) > 0.5, 1.0, -1.0) + torch.randint(-2, 3, off.shape, generator=g).double()
    off.requires_grad_(True)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    fn = lambda *a: deform_conv(*a, 1, 1, 1, 1, dg)  # noqa: E731
    assert torch.autograd.gradcheck(fn, (x, off, w), eps=1e-6, atol=1e-6, rtol=1e-5)


def test_generic_geometry_zero_offsets_is_grouped_conv2d_and_f64_finite_differences():
    """The oracle over the operator's whole geometry space (2 x 3 kernel, anisotropic stride / padding / dilation, 2 groups, 2 deformable
    groups): with zero offsets and a unit mask it is torch's grouped conv2d; its gradients pass f64 finite differences."""
    g = torch.Generator().manual_seed(7)
    B, C, Co, H, W, groups, dg = 2, 4, 6, 7, 8, 2, 2
    kh, kw, stride, pad, dil = 2, 3, (2, 1), (1, 0), (1, 2)
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C // groups, kh, kw, generator=g, dtype=torch.float64)
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    off0 = torch.zeros(B, dg * 2 * kh * kw, Ho, Wo, dtype=torch.float64)
    m1 = torch.ones(B, dg * kh * kw, Ho, Wo, dtype=torch.float64)
    out = modulated_deform_conv(x, off0, m1, w, b, stride, pad, dil, groups, dg)
    assert rel_err(out, F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil, groups=groups)) < 1e-12
    off = (torch.rand(B, dg * 2 * kh * kw, Ho, Wo, generator=g, dtype=torch.float64) - 0.5) * 1.6 + 0.05
    m = torch.rand(B, dg * kh * kw, Ho, Wo, generator=g, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (x, off, m, w, b)]

    def fn(x_, o_, m_, w_, b_):
        return modulated_deform_conv(x_, o_, m_, w_, b_, stride, pad, dil, groups, dg)
    assert torch.autograd.gradcheck(fn, leaves, eps=1e-6, atol=1e-6, rtol=1e-5)


# ---- an oracle-INDEPENDENT pin (round 6): spatially constant offsets make the operator a sum of zero-filled shifts and 1 x 1 convolutions
# (tests/dcn_composition.py: torch slicing + F.conv2d only, no code shared with the oracle).  It pins what the identities above cannot
# see with their uniform masks and group-independent offsets: the mask channel order g*K + k, the offset channel order g*2K + 2k (+1),
# the channel -> deformable-group map c // cpg, bilinear weights, the zero-outside rule incl. the (-1, 0) / (H - 1, H) strips -- for the
# forward and, through autograd on the construction, for all five gradients (grad_offset as per-(group, tap) sums over the pixels).
def _composition_case(case, kind, dtype, tol):
    import dcn_composition as DC
    B, C, Co, dg, H, W, seed = case
    x, dyx, mask, w, b, gout = DC.make_case(B, C, Co, dg, H, W, seed, dtype=dtype, kind=kind)
    ref_out, (gx, gdyx, gm, gw, gb) = DC.composition_reference(x, dyx, mask, w, b, gout, dg)
    leaves = [t.clone().requires_grad_(True) for t in (x, DC.offset_field(dyx, B, H, W), mask, w, b)]
    out = modulated_deform_conv(*leaves, 1, 1, 1, 1, dg)
    out.backward(gout)
    assert rel_err(out, ref_out) < tol, 'out'
    assert rel_err(leaves[0].grad, gx) < tol, 'grad_input'
    ok = DC.offset_grad_comparable(dyx)[:, :, None].to(gdyx.dtype)   # (all but taps that sit exactly on the -1 boundary)
    assert rel_err(DC.offset_grad_sums(leaves[1].grad, dg) * ok, gdyx * ok) < 10 * tol, 'grad_offset (summed over pixels)'
    assert rel_err(leaves[2].grad, gm) < tol, 'grad_mask'
    assert rel_err(leaves[3].grad, gw) < tol, 'grad_weight'
    assert rel_err(leaves[4].grad, gb) < tol, 'grad_bias'


COMPOSITION_CASES = [(2, 8, 6, 2, 6, 7, 11), (1, 12, 5, 3, 5, 9, 12), (2, 16, 8, 8, 7, 6, 13), (1, 6, 4, 1, 8, 5, 14)]


def test_constant_offset_composition_f64():
    for case in COMPOSITION_CASES:
        for kind in ('mixed', 'fractional', 'integer'):
            _composition_case(case, kind, torch.float64, 1e-12)


def test_constant_offset_composition_f32():
    for case in COMPOSITION_CASES:
        for kind in ('mixed', 'integer'):
            _composition_case(case, kind, torch.float32, 2e-6)
