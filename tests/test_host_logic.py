"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the header
declares, the Python mirror keeps the reference's API / state_dict schema, and nothing silently
falls back to the CPU."""
import os
import re

import pytest
import torch

from conftest import REPO, load_golden, golden_sd


def _header_symbols():
    text = open(os.path.join(REPO, 'include', 'realvsr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rvsr_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from realvsr_amd import _lib
    _lib.build()
    L = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(L, name), name
        assert name in _lib.SIGNATURES, 'ctypes signature missing for ' + name
    assert sorted(_lib.SIGNATURES) == declared


def test_operators_refuse_cpu_tensors():
    from realvsr_amd.archs.dcn import modulated_deform_conv, ModulatedDeformConvPack
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd import loss as L
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(torch.randn(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.ones(1, 9, 4, 4),
                              torch.randn(8, 8, 3, 3), None, 1, 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        ModulatedDeformConvPack(8, 8, 3, padding=1, deformable_groups=1)(torch.randn(1, 8, 4, 4))
    with pytest.raises(NotImplementedError):
        EDVR(nf=16, nframes=3, groups=4, front_RBs=1, back_RBs=1)(torch.rand(1, 3, 3, 8, 8))
    with pytest.raises(NotImplementedError):
        L.LapPyrLoss(3, 'cb', 'cb')(torch.rand(1, 1, 16, 16), torch.rand(1, 1, 16, 16))
    with pytest.raises(NotImplementedError):
        L.LapPyrLoss(3, 'ssim', 'cb')(torch.rand(1, 1, 64, 64), torch.rand(1, 1, 64, 64))
    with pytest.raises(NotImplementedError):
        L.PyramidLoss(3, 'gau', 'hb')(torch.rand(1, 1, 16, 16), torch.rand(1, 1, 16, 16))
    with pytest.raises(ValueError):
        L.LapPyrLoss(3, 'l1', 'cb')      # loss.py:205-206
    with pytest.raises(ValueError):
        L.PyramidLoss(3, 'gau', 'ssim')  # loss.py:176-177


def test_state_dict_schema_matches_reference():
    """Keys/shapes of the fixtures come from instantiating the reference modules (make_golden.py)."""
    from realvsr_amd.archs.EDVR_arch import EDVR, PCD_Align, TSA_Fusion
    from realvsr_amd.archs.dcn import ModulatedDeformConvPack
    for name, mod in [('edvr_tsa', EDVR(nf=16, nc=3, nframes=3, groups=4, front_RBs=2, back_RBs=2, w_TSA=True)),
                      ('pcd_align', PCD_Align(nf=16, groups=4)), ('tsa_fusion', TSA_Fusion(nf=16, nframes=3, center=1)),
                      ('dcn_pack', ModulatedDeformConvPack(16, 12, 3, stride=1, padding=1, dilation=1,
                                                           deformable_groups=4, extra_offset_mask=True))]:
        mod.load_state_dict(golden_sd(load_golden(name)), strict=True)
    net = EDVR(nf=64, nc=3, nframes=5, groups=8, front_RBs=5, back_RBs=10, w_TSA=True)
    assert len(net.state_dict()) == 144 and sum(p.numel() for p in net.parameters()) == 3300131
    pack = ModulatedDeformConvPack(16, 12, 3, padding=1, deformable_groups=4)
    assert float(pack.conv_offset_mask.weight.abs().sum()) == 0.0 and float(pack.bias.abs().sum()) == 0.0


def test_define_g():
    from realvsr_amd.VideoSR_archs import define_G
    from realvsr_amd.archs.EDVR_arch import EDVR_NoUp
    opt = {'network_G': {'which_model_G': 'EDVR_NoUp', 'nf': 64, 'nc': 3, 'nframes': 3, 'groups': 8, 'front_RBs': 5,
                         'back_RBs': 10, 'w_TSA': False}}
    net = define_G(opt)
    assert isinstance(net, EDVR_NoUp) and net.center == 1 and isinstance(net.tsa_fusion, torch.nn.Conv2d)
    from realvsr_amd.archs.TDAN_arch import TDAN
    tdan = define_G({'scale': 1, 'network_G': {'which_model_G': 'TDAN', 'nf': 64, 'nc': 3, 'nframes': 3, 'nb_f': 1,
                                               'nb_b': 1, 'groups': 8}})
    assert isinstance(tdan, TDAN)
    with pytest.raises(NotImplementedError):
        define_G({'network_G': {'which_model_G': 'RCAN'}})


def test_create_model_refuses_cpu_and_unknown_models():
    from realvsr_amd.VideoSR_model import create_model
    opt = {'model': 'VideoSR_AllPair_YCbCr_Split', 'gpu_ids': None, 'is_train': True, 'dist': False,
           'network_G': {'which_model_G': 'EDVR', 'nf': 16, 'nc': 3, 'nframes': 3, 'groups': 4, 'front_RBs': 1,
                         'back_RBs': 1, 'w_TSA': True}}
    with pytest.raises(NotImplementedError):
        create_model(opt)                     # no CPU path
    with pytest.raises(NotImplementedError):
        create_model(dict(opt, model='VideoSRGAN_AllPair_YCbCr_Split'))


def test_flat_buffers_layout_and_guards():
    """optim.FlatBuffers: parameters and gradients become views of two identically laid-out buffers (CPU tensors are
    fine for the layout logic); re-binding a gradient is detected."""
    from realvsr_amd.optim import FlatBuffers
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Conv2d(5, 2, 1))
    before = [p.detach().clone() for p in net.parameters()]
    fb = FlatBuffers([list(net.parameters())])
    assert fb.order[0] is list(net.parameters())[-1]          # reverse registration order
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), b)
        assert fb.offset[p] % 64 == 0
        assert p.data_ptr() == fb.param.data_ptr() + 4 * fb.offset[p]
        assert p.grad.data_ptr() == fb.grad.data_ptr() + 4 * fb.offset[p]
    net(torch.randn(2, 3, 6, 6)).sum().backward()
    assert float(fb.grad.abs().sum()) > 0                      # autograd accumulated into the views
    fb.check_bound()
    # zero_grad: one memset, p.grad = None until backward delivers (autograd then ADOPTS the incoming tensor; the fused operators
    # deliver the flat view itself, a plain torch module delivers a temporary that rebind() copies home)
    fb.zero_grad()
    assert all(p.grad is None for p in net.parameters()) and float(fb.grad.abs().sum()) == 0
    fb.check_bound()
    net(torch.randn(2, 3, 6, 6)).sum().backward()
    with pytest.raises(RuntimeError):
        fb.check_bound()                                       # adopted temporaries live outside the flat buffer ...
    want = [p.grad.clone() for p in net.parameters()]
    fb.rebind()
    fb.check_bound()                                           # ... until rebind() copies them home
    for p, w in zip(net.parameters(), want):
        assert p.grad.data_ptr() == fb.grad.data_ptr() + 4 * fb.offset[p] and torch.equal(p.grad, w)
    net.zero_grad(set_to_none=True)
    fb.rebind()
    fb.check_bound()
    # the hook the fused operators use: a parameter without a gradient gets its flat view as the output buffer
    from realvsr_amd.functional import _pgrad
    p0 = next(net.parameters())
    p0.grad = None
    assert _pgrad(p0).data_ptr() == fb.grad.data_ptr() + 4 * fb.offset[p0]
    p0.grad = torch.zeros_like(p0)
    assert _pgrad(p0).data_ptr() != fb.grad.data_ptr() + 4 * fb.offset[p0]


def test_pgrad_hands_a_home_out_once_per_step():
    """functional._pgrad: a parameter used by TWO fused operators in one backward must not get the same
    flat-buffer view twice -- AccumulateGrad has not run between the uses, p.grad is still None, and aliased outputs would turn
    g1 + g2 into 2 * g_last.  Emulated on CPU with an autograd Function that writes its weight gradient the way the fused
    operators do."""
    from realvsr_amd.functional import _pgrad
    from realvsr_amd.optim import FlatBuffers

    class Scale(torch.autograd.Function):     # y = w * x, gradient of w written into _pgrad's buffer
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x)
            ctx.w = w
            return x * w

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            gw = _pgrad(ctx.w)
            gw.copy_(g * x)
            return g * ctx.w, gw

    w = torch.nn.Parameter(torch.tensor([1., 2., 3., 4.]))
    fb = FlatBuffers([[w]])
    fb.zero_grad()
    x1, x2 = torch.tensor([1., 2., 3., 4.]), torch.tensor([10., 20., 30., 40.])
    (Scale.apply(x1, w) + Scale.apply(x2, w)).sum().backward()
    fb.rebind()
    assert torch.equal(w.grad, x1 + x2)                     # not 2 * x1 (or 2 * x2)
    assert w.grad.data_ptr() == fb.grad.data_ptr() + 4 * fb.offset[w]
    # next step: the home can be claimed again; a first use gets the flat view, the second a temporary
    fb.zero_grad()
    a, b = _pgrad(w), _pgrad(w)
    assert a.data_ptr() == fb.grad.data_ptr() + 4 * fb.offset[w] and b.data_ptr() != a.data_ptr()
    # torch.autograd.grad(): the fused backward wrote into the home but autograd handed the result to the caller, not to
    # p.grad -- rebind() must not present that as this step's gradient
    fb.zero_grad()
    (gw,) = torch.autograd.grad(Scale.apply(x1, w).sum(), [w])
    assert torch.equal(gw, x1) and w.grad is None
    fb.rebind()
    assert float(w.grad.abs().sum()) == 0


def test_gradsink_refuses_concat_convs():
    from realvsr_amd import functional as RF
    x = torch.zeros(1, 2, 4, 4)
    conv = torch.nn.Conv2d(4, 2, 3, padding=1)
    with pytest.raises((RuntimeError, NotImplementedError)) as e:
        RF.conv2d(x, conv, x2=x, sink=RF.GradSink())
    # the sink check comes first (a CPU tensor would raise NotImplementedError after it)
    assert 'GradSink' in str(e.value)


def test_gauss_kernel_gain_cache_is_not_fooled_by_address_reuse():
    from realvsr_amd import util
    k = util.gauss_kernel(channels=3)
    assert util._kernel_gain(k) == 1.0 and util._kernel_gain(4 * k) == 4.0
    other = k.clone()
    assert util._kernel_gain(other) == 1.0
    other[0, 0, 0, 0] = 7.0                                   # same object, new version: validated again
    with pytest.raises(NotImplementedError):
        util._kernel_gain(other)
    k.mul_(2.0)                                                # the tag of gauss_kernel() only holds for version 0
    assert util._kernel_gain(k) == 2.0


def test_cutblur_box_follows_python_slice_semantics():
    """with alpha near 0 the reference's cut_ratio can go negative; [cy:cy+ch] then wraps like a python slice.
    And the resolution check precedes every RNG draw (data/augments_video_allpair.py:55-56)."""
    import numpy as np
    from realvsr_amd import augment
    plan = augment.AugPlan()
    np.random.seed(3)
    state = np.random.get_state()
    with pytest.raises(ValueError):
        augment._draw_cutblur(plan, (1, 2, 3, 8, 8), 1.0, 0.7, size1=(1, 2, 3, 4, 4))
    assert all(np.array_equal(a, b) for a, b in zip(state[1:3], np.random.get_state()[1:3]))   # nothing drawn
    # brute-force the reference's arithmetic for a negative ratio and compare the box with an actual slice assignment
    for seed in range(40):
        np.random.seed(seed)
        size = (1, 2, 3, 9, 11)
        plan = augment.AugPlan()
        augment._draw_cutblur(plan, size, 1.0, 0.004)
        np.random.seed(seed)
        np.random.rand(1)
        ratio = np.random.randn() * 0.01 + 0.004
        h, w = size[2], size[3]
        ch, cw = int(h * ratio), int(w * ratio)
        cy, cx = np.random.randint(0, h - ch + 1), np.random.randint(0, w - cw + 1)
        m = np.zeros(size[-2:], bool)
        m[cy:cy + ch, cx:cx + cw] = True
        y0, y1, x0, x1 = plan.box
        m2 = np.zeros(size[-2:], bool)
        m2[y0:y1, x0:x1] = True
        assert np.array_equal(m, m2), (seed, plan.box, (cy, ch, cx, cw))


def test_flat_adam_skips_missing_grads_and_resets_state():
    """torch.optim.Adam semantics the dense flat update must keep: parameters whose grad is None do not move,
    and clearing optimizer.state (MultiStepLR_Restart(clear_state=True)) restarts the moments.  The kernel launch itself needs a
    GPU; here the arithmetic is replaced by torch's so that the host logic around it is what is tested."""
    from realvsr_amd import optim, functional as RF

    def adam_cpu(param, grad, m, v, step_size, b1, b2, eps, wd, bc2s):
        g = grad + wd * param if wd else grad
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        param.addcdiv_(m, v.sqrt() / bc2s + eps, value=-step_size)

    orig = RF.adam_step_
    RF.adam_step_ = adam_cpu
    try:
        torch.manual_seed(0)
        a, b = torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(7))
        ra, rb = torch.nn.Parameter(a.detach().clone()), torch.nn.Parameter(b.detach().clone())
        opt, ref = optim.FlatAdam([a, b], lr=1e-2, weight_decay=0.1), torch.optim.Adam([ra, rb], lr=1e-2, weight_decay=0.1)
        for it in range(4):
            opt.zero_grad()
            ref.zero_grad(set_to_none=True)
            (a * a).sum().backward()
            (ra * ra).sum().backward()
            if it != 1:                        # step 1: b gets no gradient at all
                (b * 3).sum().backward()
                (rb * 3).sum().backward()
            b_before, m_before = b.detach().clone(), opt.state[b]['exp_avg'].clone() if b in opt.state else None
            opt.step()
            ref.step()
            assert torch.allclose(a, ra, atol=1e-6), it
            if it <= 1:
                # (after a skipped step torch's PER-PARAMETER step count lags the group's; FlatAdam counts per group -- documented)
                assert torch.allclose(b, rb, atol=1e-6), it
            if it == 1:
                assert torch.equal(b.detach(), b_before) and torch.equal(opt.state[b]['exp_avg'], m_before)
            if it == 2:                        # restart: the reference's scheduler empties optimizer.state
                from collections import defaultdict
                opt.state = defaultdict(dict)
                ref.state = defaultdict(dict)
        assert float(opt.state[a]['step']) == 1.0
    finally:
        RF.adam_step_ = orig


def test_every_autograd_function_carries_the_device_guard():
    """VERDICT r2 #8 / deform_conv_cuda.cpp:499,581 (at::DeviceGuard): every fused operator's forward AND backward run under
    torch.cuda.device(<device of its tensors>) when that is not the current device.  The guard is applied to all Function
    classes of realvsr_amd.functional at import time; here: none was missed, and it switches (recorded with a stand-in context
    manager on CPU tensors that claim to live on cuda:1)."""
    import torch.autograd
    from realvsr_amd import functional as RF
    fns = [c for c in vars(RF).values() if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function]
    assert len(fns) >= 18
    for c in fns:
        assert hasattr(c.forward, '__wrapped__') and hasattr(c.backward, '__wrapped__'), c.__name__

    entered = []

    class FakeTensor:
        is_cuda = True
        device = torch.device('cuda', 1)

    class Recorder:
        def __init__(self, dev):
            self.dev = dev

        def __enter__(self):
            entered.append(self.dev)

        def __exit__(self, *a):
            return False

    import unittest.mock as mock
    with mock.patch.object(torch, 'is_tensor', lambda t: isinstance(t, FakeTensor)), \
            mock.patch.object(torch.cuda, 'current_device', lambda: 0), mock.patch.object(torch.cuda, 'device', Recorder):
        guarded = RF._guarded(lambda ctx, *a: 'ran')
        assert guarded(None, FakeTensor()) == 'ran' and entered == [torch.device('cuda', 1)]
        entered.clear()
        FakeTensor.device = torch.device('cuda', 0)          # already current: no switch
        assert guarded(None, FakeTensor()) == 'ran' and entered == []


def test_no_matrix_core_instruction_overwrites_its_own_operand():
    """hipcc may allocate the destination of an accumulator's FIRST MFMA (SrcC = 0, untied form) on top of an operand that dies at the
    instruction; on the MI355X the products of the upper operand half then come out wrong, differently from run to run (round 5,
    profiles/r05_notes.md; bf16x3.h: mfma_bf16_first).  tools/check_mfma_overlap.py compiles a kernel file to gfx950 ISA and looks for
    the pattern: run here on the two files whose kernels start accumulators that way (no GPU needed; ~1 minute)."""
    import shutil
    import subprocess
    import sys
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not on PATH')
    csrc = os.path.join(REPO, 'realvsr_amd', 'csrc')
    out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'check_mfma_overlap.py'), os.path.join(csrc, 'dcn5_kernels.hip'),
                          os.path.join(csrc, 'dcn6_kernels.hip')], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '0 overlapping' in out.stdout
