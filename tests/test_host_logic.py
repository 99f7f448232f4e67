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
