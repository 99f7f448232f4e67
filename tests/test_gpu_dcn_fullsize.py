"""Full-resolution parity of the modulated deformable convolution against the CPU oracle at realistic offsets.  -m gpu

tests/test_gpu_dcn.py covers the operator on small frames (<= 45 x 80) and tests/test_gpu_fullsize.py proves the tile seams
of a 180 x 320 frame only at zero / integer offsets.  Here the operator itself -- forward and all five gradients, through the
C ABI, in both GEMM modes -- meets the oracle (oracle/dcn_oracle.c, a line map of kernel.cu:467-767) at BASELINE's frame sizes
and at offset standard deviations of 0.1 / 3 / 6 px: every device-selected backward kernel, the halo logic of the LDS tiles
at real seam geometry (23 x 10 workgroup tiles per frame) and the out-of-tile global gather of the forward
(dcn3_kernels.hip) are exercised against reference arithmetic, not against properties.  The oracle needs ~3 s per case on the
GPU box's host cores; its results are shared between the two GEMM modes.
"""
import functools

import pytest
import torch

from gpu_util import check, dev, gemm_modes

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu


def _inputs(B, C, H, W, ostd, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 8 * 18, H, W, generator=g) * ostd
    m = torch.rand(B, 8 * 9, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(C, generator=g)
    gout = torch.randn(B, C, H, W, generator=g)
    return x, off, m, w, b, gout


@functools.lru_cache(maxsize=8)   # (both GEMM modes share an oracle result: ~1 GB of host memory in total)
def _oracle(B, C, H, W, ostd, seed, backward):
    from oracle.dcn_oracle import modulated_deform_conv
    x, off, m, w, b, gout = _inputs(B, C, H, W, ostd, seed)
    leaves = [t.requires_grad_(backward) for t in (x, off, m, w, b)]
    out = modulated_deform_conv(*leaves, 1, 1, 1, 1, 8)
    if backward:
        out.backward(gout)
    return out.detach(), [l.grad for l in leaves]


CASES = [(2, 64, 180, 320, 0.1), (2, 64, 180, 320, 3.0), (2, 64, 180, 320, 6.0),
         (1, 128, 180, 320, 0.1), (1, 128, 180, 320, 3.0), (1, 128, 180, 320, 6.0)]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B%d-C%d-%dx%d-std%g' % c)
def test_fullsize_forward_and_gradients_vs_oracle(case, gemm_mode):
    from realvsr_amd.archs.dcn import modulated_deform_conv
    B, C, H, W, ostd = case
    seed = int(1000 * ostd) + C
    oref, gref = _oracle(B, C, H, W, ostd, seed, True)
    d = dev()
    x, off, m, w, b, gout = _inputs(B, C, H, W, ostd, seed)
    leaves = [t.to(d).requires_grad_(True) for t in (x, off, m, w, b)]
    out = modulated_deform_conv(*leaves, 1, 1, 1, 1, 8)
    out.backward(gout.to(d))
    torch.cuda.synchronize()
    # tolerances of tests/test_gpu_dcn.py: exact-f32 MFMA 2e-5, bf16x3 1e-4 (relative to the tensor's max); the atomic scatter
    # of grad_input and the 57 600-pixel reductions of grad_weight / grad_bias get the looser of the two in both modes
    check('out', out, oref, 2e-5 if gemm_mode == 'f32' else 1e-4)
    for name, a, r in zip(('grad_input', 'grad_offset', 'grad_mask', 'grad_weight', 'grad_bias'), [l.grad for l in leaves], gref):
        check(name, a, r, 1e-4)


def test_config5_frame_forward_vs_oracle(gemm_mode):
    """Forward only, 128 channels at the 540 x 960 LR frame of BASELINE config 5 (68 x 30 workgroup tiles, 3.2 px offsets)."""
    from realvsr_amd.archs.dcn import modulated_deform_conv
    B, C, H, W, ostd, seed = 1, 128, 540, 960, 4.0, 77
    oref, _ = _oracle(B, C, H, W, ostd, seed, False)
    d = dev()
    x, off, m, w, b, _ = _inputs(B, C, H, W, ostd, seed)
    with torch.no_grad():
        out = modulated_deform_conv(x.to(d), off.to(d), m.to(d), w.to(d), b.to(d), 1, 1, 1, 1, 8)
    torch.cuda.synchronize()
    check('out', out, oref, 2e-5 if gemm_mode == 'f32' else 1e-4)


@pytest.mark.parametrize('ostd,halo', [(0.1, 3), (2.5, 7), (6.0, 11)])   # (std 2.5: 16 % beyond 3.5 px, 0.3 % beyond 7.5 px)
def test_fused_pack_halo_selection_vs_oracle(ostd, halo, gemm_mode):
    """The fused pack (functional.dcn_pack, raw conv_offset_mask tensor) picks the forward's LDS tile halo itself: in training from
    the offset counters the PREVIOUS backward of the layer left on the host (functional.DcnOffsetStats), without gradients from a
    probe pass on the device.  Both paths against the oracle on a 180 x 320 frame: pass 1 has no statistic yet (3 px tile), pass 2
    runs the tile the counters ask for, the no-grad call selects on the device; all three must agree with the oracle, and the
    recorded decision must be the one the offsets imply."""
    from realvsr_amd import functional as RF
    B, C, H, W, seed = 1, 64, 180, 320, int(1000 * ostd) + 3
    x, off, m, w, b, gout = _inputs(B, C, H, W, ostd, seed)
    logit = torch.log(m.clamp(1e-4, 1 - 1e-4)) - torch.log1p(-m.clamp(1e-4, 1 - 1e-4))
    om = torch.cat([off, logit], 1)
    from oracle.dcn_oracle import modulated_deform_conv
    ref_leaves = [t.clone().requires_grad_(True) for t in (x, om, w, b)]
    o_, m_ = ref_leaves[1][:, :144], torch.sigmoid(ref_leaves[1][:, 144:])
    oref = modulated_deform_conv(ref_leaves[0], o_, m_, ref_leaves[2], ref_leaves[3], 1, 1, 1, 1, 8)
    oref.backward(gout)
    d = dev()
    wd = w.to(d).requires_grad_(True)
    tol = 2e-5 if gemm_mode == 'f32' else 1e-4
    for it in range(2):
        leaves = [x.to(d).requires_grad_(True), om.to(d).requires_grad_(True), wd, b.to(d).requires_grad_(True)]
        wd.grad = None
        out = RF.dcn_pack(*leaves, 1, 1, 1, 8)
        out.backward(gout.to(d))
        torch.cuda.synchronize()
        check('out pass %d' % it, out, oref.detach(), tol)
        for name, a, r in zip(('grad_input', 'grad_offset_mask', 'grad_weight', 'grad_bias'), [l.grad for l in leaves], [l.grad for l in ref_leaves]):
            check('%s pass %d' % (name, it), a, r, 1e-4)
    if gemm_mode != 'f32':   # (the exact-f32 mode runs the first-generation kernels: no tile halo to choose)
        assert RF.dcn_offset_stats.forward_halo(wd, C) == halo
    with torch.no_grad():
        out = RF.dcn_pack(x.to(d), om.to(d), wd.detach(), b.to(d), 1, 1, 1, 8)
    torch.cuda.synchronize()
    check('out no-grad', out, oref.detach(), tol)
