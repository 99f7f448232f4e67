"""Pins oracle/edvr_oracle.py against fixtures produced by the IMPORTED reference modules
(tests/golden/make_golden.py); CPU only."""
import numpy as np
import torch

from conftest import load_golden, golden_sd, rel_err
from oracle import edvr_oracle as O


def _t(g, k, grad=False):
    return torch.from_numpy(g[k]).clone().requires_grad_(grad)


def test_dcn_pack_fixture():
    g = load_golden('dcn_pack')
    sd = {('p.' + k): v.requires_grad_(True) for k, v in golden_sd(g).items()}
    x, feat = _t(g, 'x', True), _t(g, 'feat', True)
    out = O.dcn_pack(sd, 'p', x, feat, 4)
    out.backward(_t(g, 'gout'))
    assert rel_err(out, g['out']) < 1e-6
    assert rel_err(x.grad, g['gx']) < 1e-5 and rel_err(feat.grad, g['gfeat']) < 1e-5
    for k, v in sd.items():
        assert rel_err(v.grad, g['grad.' + k[2:]]) < 1e-5, k


def test_pcd_align_fixture():
    g = load_golden('pcd_align')
    sd = {('pcd.' + k): v.requires_grad_(True) for k, v in golden_sd(g).items()}
    nbr = [_t(g, 'nbr%d' % l, True) for l in range(3)]
    ref = [_t(g, 'ref%d' % l, True) for l in range(3)]
    out = O.pcd_align(sd, 'pcd', nbr, ref, int(g['groups']))
    out.backward(_t(g, 'gout'))
    assert rel_err(out, g['out']) < 1e-5
    for l in range(3):
        assert rel_err(nbr[l].grad, g['gnbr%d' % l]) < 1e-4
        assert rel_err(ref[l].grad, g['gref%d' % l]) < 1e-4
    for k, v in sd.items():
        assert rel_err(v.grad, g['grad.' + k[4:]]) < 1e-4, k


def test_tsa_fusion_fixture():
    g = load_golden('tsa_fusion')
    sd = {('tsa.' + k): v.requires_grad_(True) for k, v in golden_sd(g).items()}
    al = _t(g, 'aligned', True)
    out = O.tsa_fusion(sd, 'tsa', al, 1)
    out.backward(_t(g, 'gout'))
    assert rel_err(out, g['out']) < 1e-5
    assert rel_err(al.grad, g['galigned']) < 1e-4
    for k, v in sd.items():
        assert rel_err(v.grad, g['grad.' + k[4:]]) < 1e-4, k


def _edvr_loss(out, gt):
    return O.lap_pyr_loss(out[:, 0:1], gt[:, 0:1], 3) + O.charbonnier(out[:, 1:3], gt[:, 1:3])


def test_edvr_tsa_fixture():
    g = load_golden('edvr_tsa')
    sd = {k: v.requires_grad_(True) for k, v in golden_sd(g).items()}
    out = O.edvr_forward(sd, _t(g, 'x'), nframes=3, groups=4, front_RBs=2, back_RBs=2, w_TSA=True)
    assert rel_err(out, g['out']) < 1e-5
    loss = _edvr_loss(out, _t(g, 'gt'))
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-6 * abs(float(g['loss']))
    gnorm = torch.sqrt(sum((v.grad ** 2).sum() for v in sd.values())).item()
    assert abs(gnorm - float(g['gnorm'])) < 1e-4 * float(g['gnorm'])
    for k, v in sd.items():
        assert rel_err(v.grad, g['grad.' + k]) < 2e-4, k


def test_edvr_noup_fixture():
    from weights import fill_state_dict
    import types
    g = load_golden('edvr_noup')
    # shapes of the reference-schema state_dict come from the product module definition
    from realvsr_amd.archs import EDVR_arch
    net = EDVR_arch.EDVR_NoUp(nf=64, nc=3, nframes=3, groups=8, front_RBs=1, back_RBs=1, w_TSA=False)
    fill_state_dict(net, 77)
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    out = O.edvr_forward(sd, _t(g, 'x'), nframes=3, groups=8, front_RBs=1, back_RBs=1, w_TSA=False,
                         upscale=False)
    assert rel_err(out, g['out']) < 1e-5
    loss = _edvr_loss(out, _t(g, 'gt'))
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-6 * abs(float(g['loss']))
    for k in [k for k in g if k.startswith('grad.')]:
        assert rel_err(sd[k[5:]].grad, g[k]) < 2e-4, k


def test_pyramids_bit_exact_on_integer_images():
    g = load_golden('pyramid_int')
    for tag in 'abc':
        img = _t(g, 'img_' + tag)
        for name, fn, lv in (('laplacian', O.laplacian_pyramid, 3), ('lap', O.lap_pyramid, 2),
                             ('gau', O.gau_pyramid, 3)):
            for i, level in enumerate(fn(img, lv)):
                assert np.array_equal(level.numpy(), g['%s_%s_%d' % (name, tag, i)]), (name, tag, i)


def test_losses_fixture():
    g = load_golden('losses')
    fns = {'lappyr_cb': lambda x, y: O.lap_pyr_loss(x, y, 3),
           'lappyr_cb_sum': lambda x, y: O.lap_pyr_loss(x, y, 2, 'sum'),
           'pyr_gau_cb': lambda x, y: O.pyramid_loss(x, y, 3, 'gau', 'cb'),
           'pyr_lap_l1': lambda x, y: O.pyramid_loss(x, y, 2, 'lap', 'l1'),
           'pyr_gau_l2': lambda x, y: O.pyramid_loss(x, y, 3, 'gau', 'l2'),
           'cb': lambda x, y: O.charbonnier(x, y), 'gw': lambda x, y: O.gw_loss(x, y, 4),
           'gw_sum': lambda x, y: O.gw_loss(x, y, 2, 'sum')}
    for tag in ('y', 'rgb'):
        for name, fn in fns.items():
            x = _t(g, 'x_' + tag, True)
            l = fn(x, _t(g, 'y_' + tag))
            l.backward()
            ref = float(g['%s_%s' % (name, tag)])
            assert abs(l.item() - ref) < 1e-6 * abs(ref), (name, tag)
            assert rel_err(x.grad, g['g_%s_%s' % (name, tag)]) < 1e-5, (name, tag)
