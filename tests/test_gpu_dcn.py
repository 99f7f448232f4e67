"""HIP modulated deformable convolution vs the CPU oracle + committed fixtures.  -m gpu"""
import pytest
import torch

from conftest import load_golden, golden_sd
from gpu_util import check, check_l2, dev, gemm_modes

gemm_mode = gemm_modes()

pytestmark = pytest.mark.gpu
TOL = 2e-5
TOL_G = 1e-4  # grad_input is an atomic scatter (order-nondeterministic, like the reference)


def _run_hip(x, off, m, w, b, stride, pad, dil, dg, gout):
    from realvsr_amd.archs.dcn import modulated_deform_conv
    d = dev()
    leaves = [t.to(d).requires_grad_(True) if t is not None else None for t in (x, off, m, w, b)]
    out = modulated_deform_conv(*leaves, stride, pad, dil, 1, dg)
    out.backward(gout.to(d))
    torch.cuda.synchronize()
    return out, [l.grad if l is not None else None for l in leaves]


def _run_oracle(x, off, m, w, b, stride, pad, dil, dg, gout):
    from oracle.dcn_oracle import modulated_deform_conv
    leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (x, off, m, w, b)]
    out = modulated_deform_conv(*leaves, stride, pad, dil, 1, dg)
    out.backward(gout)
    return out, [l.grad if l is not None else None for l in leaves]


def _compare(args, gout, tol=TOL):
    out, grads = _run_hip(*args, gout)
    oref, gref = _run_oracle(*args, gout)
    check('out', out, oref, tol)
    for name, a, r in zip(('grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), grads, gref):
        if r is not None:
            check(name, a, r, max(TOL_G, tol))   # (the speed modes pass their own tolerance)


def test_fixture_dcn_op(gemm_mode):
    TOL = 2e-5 if gemm_mode == 'f32' else 1e-4
    g = load_golden('dcn_op')
    t = {k: torch.from_numpy(v) for k, v in g.items() if v.ndim > 0}
    out, grads = _run_hip(t['x'], t['offset'], t['mask'], t['weight'], t['bias'], 1, 1, 1, int(g['dg']), t['gout'])
    check('out', out, t['out'], TOL)
    for name, a in zip(('gx', 'goffset', 'gmask', 'gweight', 'gbias'), grads):
        check(name, a, t[name], TOL_G)


SHAPES = [
    # B, C, Co, dg, H, W, stride, pad, dil, offset_std, bias
    (2, 64, 64, 8, 20, 36, 1, 1, 1, 2.0, True),     # EDVR L1-like, cpg 8
    (1, 128, 128, 8, 12, 40, 1, 1, 1, 1.0, True),   # nf128, cpg 16 (two chunks per group)
    (2, 16, 12, 4, 7, 9, 1, 1, 1, 3.0, True),       # cpg 4 (two groups per chunk), ragged tile
    (1, 64, 64, 8, 45, 80, 1, 1, 1, 10.0, True),    # large motion, L3 size of 180x320
    (1, 32, 40, 4, 9, 33, 1, 1, 1, 1.0, False),     # no bias, Co not multiple of 32
    (1, 16, 16, 2, 11, 13, 2, 1, 1, 1.0, True),     # stride 2
    (1, 16, 16, 2, 11, 13, 1, 2, 2, 1.0, True),     # dilation 2
    (1, 8, 72, 1, 6, 34, 1, 1, 1, 1.0, True),       # Co > 64 (MT=4 path), dg 1
    (1, 64, 80, 8, 10, 36, 1, 1, 1, 0.5, True),     # backward in two output-channel passes (64 + 16)
    # the backward picks its window halo on the device from the offsets (dcn6_kernels.hip, rvsr_launch_dcn_bwdin6: 2 / 4 / 5 / 8 / 12 px):
    (1, 64, 64, 8, 24, 40, 1, 1, 1, 3.0, True),     # 40 % of the components beyond 2.5 px: the 5 px window
    (1, 64, 64, 8, 24, 40, 1, 1, 1, 6.0, True),     # 68 %: the 12 px window
    (1, 128, 128, 8, 17, 40, 1, 1, 1, 8.0, True),   # Co > 64 (NK = 8): windows 2 / 4 / 5 / 8 px only
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: '-'.join(str(v) for v in s))
def test_random_shapes_vs_oracle(shape, gemm_mode):
    B, C, Co, dg, H, W, stride, pad, dil, ostd, with_bias = shape
    g = torch.Generator().manual_seed(sum(int(v) for v in shape[:9]))
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, Ho, Wo, generator=g) * ostd
    m = torch.rand(B, dg * 9, Ho, Wo, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g) if with_bias else None
    gout = torch.randn(B, Co, Ho, Wo, generator=g)
    _compare((x, off, m, w, b, stride, pad, dil, dg), gout, {'f32': 2e-5, 'bf16x3': 1e-4}.get(gemm_mode, 2e-2))   # (speed modes: test_gpu_modes.py)


def _random_shapes(n, seed):
    """Seeded sweep: channel / group counts on both sides of the 8-channel octet, ragged tiles, every offset regime of the
    device-side backward selection (private windows, 3 px and 5 px halo, global fallback)."""
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        dg = rnd.choice([1, 2, 4, 8])
        C = dg * rnd.choice([4, 8, 8, 16])
        Co = rnd.choice([8, 24, 64, 72, 128])
        out.append((rnd.choice([1, 2]), C, Co, dg, rnd.choice([6, 9, 17, 24]), rnd.choice([8, 20, 33, 44]), 1, 1, 1,
                    rnd.choice([0.3, 1.0, 2.5, 4.0, 9.0]), rnd.random() < 0.8))
    return out


@pytest.mark.parametrize('shape', _random_shapes(14, 928), ids=lambda s: '-'.join(str(v) for v in s))
def test_random_sweep_vs_oracle(shape, gemm_mode):
    test_random_shapes_vs_oracle(shape, gemm_mode)


def test_identities(gemm_mode):
    TOL = 2e-5 if gemm_mode == 'f32' else 1e-4
    import torch.nn.functional as F
    from realvsr_amd.archs.dcn import modulated_deform_conv
    d = dev()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 16, 10, 34, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    b = torch.randn(16, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    out = modulated_deform_conv(x.to(d), torch.zeros(2, 36, 10, 34, device=d), torch.ones(2, 18, 10, 34, device=d),
                                w.to(d), b.to(d), 1, 1, 1, 1, 2)
    check('zero offset, unit mask == conv2d', out, ref, TOL)
    out = modulated_deform_conv(x.to(d), torch.full((2, 36, 10, 34), 100.0, device=d),
                                torch.ones(2, 18, 10, 34, device=d), w.to(d), b.to(d), 1, 1, 1, 1, 2)
    check('far offsets == bias', out, b.view(1, -1, 1, 1).expand(2, 16, 10, 34), 0.0)


# ---- oracle-independent pin of the HIP operator (round 6): spatially constant offsets, different per (deformable group, tap), per-pixel random
# mask: the operator is then a sum of zero-filled shifts and 1 x 1 convolutions (tests/dcn_composition.py: torch slicing + F.conv2d, float64
# autograd; no code shared with oracle/).  Pins the offset / mask channel orders, c // cpg, bilinear weights and the zero-outside rule of the
# forward and of all five gradients (grad_offset as per-(group, tap) sums) -- see tests/test_oracle_dcn.py for the same check on the oracle.
COMPOSITION_CASES = [
    # B, C, Co, dg, H, W, seed
    (2, 64, 64, 8, 20, 36, 21),     # cpg 8: dcn_fwd3 / dcn_bwdin6 / dcn_bwdw6
    (1, 128, 64, 8, 12, 40, 22),    # cpg 16 (two chunks per deformable group)
    (2, 16, 12, 4, 7, 9, 23),       # cpg 4 (two groups per chunk): the fifth-generation backward
    (1, 8, 6, 1, 9, 33, 24),        # one deformable group
]


def _composition(case, kind, dtype, tol, tol_g):
    import dcn_composition as DC
    from realvsr_amd.archs.dcn import modulated_deform_conv
    B, C, Co, dg, H, W, seed = case
    x, dyx, mask, w, b, gout = DC.make_case(B, C, Co, dg, H, W, seed, dtype=dtype, kind=kind)
    ref_out, (gx, gdyx, gm, gw, gb) = DC.composition_reference(x, dyx, mask, w, b, gout, dg)
    d = dev()
    leaves = [t.to(d).requires_grad_(True) for t in (x, DC.offset_field(dyx, B, H, W), mask, w, b)]
    out = modulated_deform_conv(*leaves, 1, 1, 1, 1, dg)
    out.backward(gout.to(d))
    torch.cuda.synchronize()
    tag = '%s %s ' % ('-'.join(str(v) for v in case[:6]), kind)
    check(tag + 'out', out, ref_out, tol)
    check(tag + 'grad_input', leaves[0].grad, gx, tol_g)
    ok = DC.offset_grad_comparable(dyx)[:, :, None].double()
    check(tag + 'grad_offset (pixel sums)', DC.offset_grad_sums(leaves[1].grad.cpu(), dg) * ok, gdyx * ok, 4 * tol_g)
    check(tag + 'grad_mask', leaves[2].grad, gm, tol_g)
    check(tag + 'grad_weight', leaves[3].grad, gw, tol_g)
    check(tag + 'grad_bias', leaves[4].grad, gb, tol_g)


@pytest.mark.parametrize('case', COMPOSITION_CASES, ids=lambda s: '-'.join(str(v) for v in s))
def test_constant_offset_composition(case, gemm_mode):
    tol = 2e-5 if gemm_mode == 'f32' else 1e-4
    for kind in ('mixed', 'fractional', 'integer'):
        _composition(case, kind, torch.float32, tol, max(TOL_G, tol))


@pytest.mark.parametrize('case', COMPOSITION_CASES[2:], ids=lambda s: '-'.join(str(v) for v in s))
def test_constant_offset_composition_f64_general_path(case):
    """float64 tensors take the operator's general path (csrc/dcn_generic.hip): agreement to double precision."""
    for kind in ('mixed', 'integer'):
        _composition(case, kind, torch.float64, 1e-11, 1e-11)


# ---- run-to-run determinism of everything but grad_input (round 6; the weight gradient comes from dcn_bwdw6: per-stream partials reduced in a fixed order)
def test_dcn_weight_grad_is_deterministic():
    import determinism_check as DET
    for case in DET.CASES:
        bad = DET.run_case(case, repeats=20)
        assert not bad, (case, bad[:8])


def test_dcn_weight_grad_is_deterministic_two_workgroups_per_cu():
    """dcn_bwdw6 as two 4-wave workgroups per CU (RVSR_BWDW6_WG=2, read once per process): the schedule round 5 could not make reproducible."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, 'determinism_check.py')], env=dict(os.environ, RVSR_BWDW6_WG='2'),
                         capture_output=True, text=True, timeout=900)
    print(out.stdout[-1500:])
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])


def test_pack_fixture(gemm_mode):
    """ModulatedDeformConvPack (fused conv_offset_mask + DCN) vs the reference wiring fixture.
    The offsets come out of a conv block, so in bf16x3 mode they carry ~1e-5 px of noise."""
    from realvsr_amd.archs.dcn import ModulatedDeformConvPack
    TOL, TOL_G = (2e-5, 1e-4) if gemm_mode == 'f32' else (5e-4, 2e-3)
    g = load_golden('dcn_pack')
    d = dev()
    pack = ModulatedDeformConvPack(16, 12, 3, stride=1, padding=1, dilation=1, deformable_groups=4,
                                   extra_offset_mask=True)
    pack.load_state_dict(golden_sd(g), strict=True)
    pack = pack.to(d)
    x = torch.from_numpy(g['x']).to(d).requires_grad_(True)
    feat = torch.from_numpy(g['feat']).to(d).requires_grad_(True)
    out = pack([x, feat])
    out.backward(torch.from_numpy(g['gout']).to(d))
    torch.cuda.synchronize()
    check('out', out, torch.from_numpy(g['out']), TOL)
    check('gx', x.grad, torch.from_numpy(g['gx']), TOL_G)
    check('gfeat', feat.grad, torch.from_numpy(g['gfeat']), TOL_G)
    for k, p in pack.named_parameters():
        check('grad.' + k, p.grad, torch.from_numpy(g['grad.' + k]), TOL_G)


def test_errors():
    from realvsr_amd.archs.dcn import modulated_deform_conv, deform_conv
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(torch.randn(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.ones(1, 9, 4, 4),
                              torch.randn(8, 8, 3, 3), None, 1, 1, 1, 1, 1)
    d = dev()
    # a 5 x 5 kernel: the operator's general path (tests/test_gpu_dcn_generic.py); zero offsets and a unit mask make it the plain convolution
    x5, w5 = torch.randn(1, 8, 8, 8, device=d), torch.randn(8, 8, 5, 5, device=d)
    out5 = modulated_deform_conv(x5, torch.zeros(1, 50, 8, 8, device=d), torch.ones(1, 25, 8, 8, device=d), w5, None, 1, 2, 1, 1, 1)
    check('5x5, zero offsets == conv2d', out5, torch.nn.functional.conv2d(x5.double().cpu(), w5.double().cpu(), None, padding=2), 2e-5)
    with pytest.raises(RuntimeError):  # an offset tensor of the wrong shape: loud, not silent
        modulated_deform_conv(x5, torch.zeros(1, 18, 8, 8, device=d), torch.ones(1, 25, 8, 8, device=d), w5, None, 1, 2, 1, 1, 1)
    with pytest.raises(RuntimeError):  # non-contiguous input, as deform_conv_cuda.cpp:497
        modulated_deform_conv(torch.randn(1, 8, 8, 16, device=d)[:, :, :, ::2], torch.zeros(1, 18, 8, 8, device=d),
                              torch.ones(1, 9, 8, 8, device=d), torch.randn(8, 8, 3, 3, device=d), None, 1, 1, 1, 1, 1)
    with pytest.raises(ValueError):   # (DCNv1 has its own file: tests/test_gpu_dcn_v1.py)
        deform_conv(torch.randn(8, 4, 4), None, None)


@pytest.mark.parametrize('cfg', [(2, 32, 48, 4, 2), (1, 32, 32, 1, 2), (1, 64, 32, 2, 4)], ids=lambda c: 'B%d-C%d-Co%d-dg%d-G%d' % c)
def test_groups_vs_oracle(cfg, gemm_mode):
    """groups > 1 (deform_conv_cuda.cpp:539-561 per-group GEMMs): composed from groups == 1 calls on channel slices."""
    from oracle.dcn_oracle import modulated_deform_conv as oracle_dcn
    from realvsr_amd.archs.dcn import modulated_deform_conv, ModulatedDeformConvPack
    B, C, Co, dg, G = cfg
    g = torch.Generator().manual_seed(C + Co + dg + G)
    H, W = 10, 36
    t = [torch.randn(B, C, H, W, generator=g), torch.randn(B, dg * 18, H, W, generator=g) * 1.5, torch.rand(B, dg * 9, H, W, generator=g),
         torch.randn(Co, C // G, 3, 3, generator=g) / (3 * (C // G) ** 0.5), torch.randn(Co, generator=g)]
    gout = torch.randn(B, Co, H, W, generator=g)
    ref = [x.clone().requires_grad_(True) for x in t]
    oref = oracle_dcn(*ref, 1, 1, 1, G, dg)
    oref.backward(gout)
    d = dev()
    got = [x.to(d).requires_grad_(True) for x in t]
    out = modulated_deform_conv(*got, 1, 1, 1, G, dg)
    out.backward(gout.to(d))
    torch.cuda.synchronize()
    check('out', out, oref.detach(), 2e-5 if gemm_mode == 'f32' else 1e-4)
    for name, a, r in zip(('grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), got, ref):
        check(name, a.grad, r.grad, 1e-4)
    # the Pack module with groups > 1: unfused wiring on the composed operator; at init (zero conv_offset_mask) = 0.5 * grouped conv + bias
    import torch.nn.functional as F
    pack = ModulatedDeformConvPack(C, Co, 3, stride=1, padding=1, dilation=1, groups=G, deformable_groups=dg).to(d)
    with torch.no_grad():
        pack.bias.copy_(t[4])
    x = t[0].to(d)
    want = 0.5 * F.conv2d(t[0].double(), pack.weight.detach().double().cpu(), None, padding=1, groups=G) + t[4].double().view(1, -1, 1, 1)
    check('pack at init', pack(x), want, 2e-5 if gemm_mode == 'f32' else 1e-4)
