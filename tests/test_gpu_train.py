"""The operators either side of netG in a training step, and the step itself.  -m gpu

  * SSIM / Huber / L1 / L2 criteria and conv_gauss / upsample vs oracle + reference-import fixtures (losses2.npz)
  * clip augmentation kernel vs the reference's outputs under seeds (augment.npz)
  * FlatAdam vs torch.optim.Adam
  * VideoSRModel.optimize_parameters vs the reference's own VideoSRModel (train_step.npz): per-step losses, gradient
    norm and parameters after 3 steps, PSNR-Y of the post-step output
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import check, check_l2, dev, gemm_modes

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev()).requires_grad_(grad)


# ------------------------------------------------------------------------------------------ criteria
def test_ssim_loss_vs_oracle():
    """HIP SSIM (fwd + both gradients) vs oracle/ssim_oracle.py -- third-party algorithm, PARITY UNPINNED."""
    from oracle import ssim_oracle as S
    from realvsr_amd import functional as RF
    from realvsr_amd.loss import SSIM
    g = torch.Generator().manual_seed(3)
    for shape in [(2, 1, 24, 32), (1, 3, 11, 11), (3, 1, 45, 80), (1, 1, 37, 19)]:
        x = torch.rand(shape, generator=g)
        y = (x + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1) if shape[-1] != 19 else 1 - x   # last: relu(cs) clamps
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        lr = S.ssim_loss(xr, yr)
        lr.backward()
        xg, yg = _t(x.numpy(), True), _t(y.numpy(), True)
        lg = RF.ssim_loss(xg, yg)
        lg.backward()
        assert abs(lg.item() - lr.item()) <= 2e-6 + 1e-5 * abs(lr.item()), (shape, lg.item(), lr.item())
        check('ssim gx %s' % (shape,), xg.grad, xr.grad, 2e-4)
        check('ssim gy %s' % (shape,), yg.grad, yr.grad, 2e-4)
    m = SSIM(channels=1)
    x, y = torch.rand(2, 1, 20, 20, generator=g), torch.rand(2, 1, 20, 20, generator=g)
    score = m(_t(x.numpy()), _t(y.numpy()), as_loss=False)
    check('ssim score', score, S.SSIM(1)(x, y, as_loss=False), 1e-5)
    with pytest.raises(RuntimeError):
        RF.ssim_loss(_t(np.zeros((1, 1, 8, 40), np.float32)), _t(np.zeros((1, 1, 8, 40), np.float32)))


def test_losses2_fixture():
    """HuberLoss, PyramidLoss('hb'), LapPyrLoss(3,'ssim','cb') (SSIM term unpinned) vs the reference's loss.py."""
    from realvsr_amd import loss as L
    g = load_golden('losses2')
    y = _t(g['y'])
    for lname, crit in [('hb', L.HuberLoss()), ('hb_sum', L.HuberLoss(delta=0.05, reduction='sum')),
                        ('pyr_gau_hb', L.PyramidLoss(3, 'gau', 'hb', 'mean')),
                        ('pyr_lap_hb', L.PyramidLoss(2, 'lap', 'hb', 'mean')),
                        ('lappyr_ssim_UNPINNED', L.LapPyrLoss(3, 'ssim', 'cb', 'mean'))]:
        x = _t(g['x'], True)
        l = crit(x, y)
        l.backward()
        assert abs(l.item() - float(g[lname])) <= 2e-5 * abs(float(g[lname])) + 1e-9, (lname, l.item(), float(g[lname]))
        check('g_' + lname, x.grad, torch.from_numpy(g['g_' + lname]), 2e-4 if 'ssim' in lname else 2e-5)


def test_l1_l2_vs_torch():
    from realvsr_amd import loss as L
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(3, 2, 17, 23, generator=g), torch.randn(3, 2, 17, 23, generator=g)
    for ours, theirs in [(L.L1Loss(), torch.nn.L1Loss()), (L.MSELoss(), torch.nn.MSELoss()),
                         (L.L1Loss('sum'), torch.nn.L1Loss(reduction='sum')), (L.MSELoss('sum'), torch.nn.MSELoss(reduction='sum'))]:
        xr = x.clone().requires_grad_(True)
        lr = theirs(xr, y)
        lr.backward()
        xg, yg = _t(x.numpy(), True), _t(y.numpy(), True)
        lg = ours(xg, yg)
        lg.backward()
        assert abs(lg.item() - lr.item()) <= 2e-6 * abs(lr.item())
        check('gx', xg.grad, xr.grad, 1e-6)
        check('gy', yg.grad, -xr.grad, 1e-6)


def test_conv_gauss_and_upsample_fixture():
    """utils/util.py:503-516 as stand-alone operators: values + gradients, bit-exact on integer images."""
    from realvsr_amd import util
    g = load_golden('losses2')
    for tag in ('a', 'b'):
        C = g['conv_gauss_%s.in' % tag].shape[1]
        k = util.gauss_kernel(channels=C, device=dev())
        for fname, fn in (('conv_gauss', lambda t: util.conv_gauss(t, k)), ('conv_gauss4', lambda t: util.conv_gauss(t, 4 * k)),
                          ('upsample', util.upsample)):
            x = _t(g['%s_%s.in' % (fname, tag)], True)
            out = fn(x)
            out.backward(_t(g['%s_%s.gout' % (fname, tag)]))
            check('%s_%s out' % (fname, tag), out, torch.from_numpy(g['%s_%s.out' % (fname, tag)]), 1e-6)
            check('%s_%s gin' % (fname, tag), x.grad, torch.from_numpy(g['%s_%s.gin' % (fname, tag)]), 2e-6)
        ii = _t(g['int_%s.in' % tag])
        assert torch.equal(util.conv_gauss(ii, k).cpu(), torch.from_numpy(g['int_%s.conv_gauss' % tag]))
        assert torch.equal(util.upsample(ii).cpu(), torch.from_numpy(g['int_%s.upsample' % tag]))
    with pytest.raises(NotImplementedError):
        util.conv_gauss(_t(g['int_a.in']), torch.ones(3, 1, 5, 5, device=dev()))
    # laplacian level == current - upsample(downsample(conv_gauss(current))) composed from the stand-alone operators
    img = _t(g['conv_gauss_a.in'])
    lap = util.laplacian_pyramid(img, None, 2)
    again = img - util.upsample(util.downsample(util.conv_gauss(img)))
    check('laplacian level from parts', again, lap[0].cpu(), 1e-6)


# ------------------------------------------------------------------------------------------ augmentation
def test_augment_kernel_matches_reference_under_seeds():
    """data/augments_video_allpair.py outputs (augment.npz) reproduced by draw_plan + ONE kernel launch."""
    from realvsr_amd import augment
    from realvsr_amd import functional as RF
    g = load_golden('augment')
    for seed in range(6):
        np.random.seed(100 + seed)
        o1, o2 = augment.cutblur(_t(g['a4']), _t(g['b4']), prob=1.0, alpha=0.7)
        assert np.array_equal(o1.cpu().numpy(), g['cutblur%d.1' % seed]) and np.array_equal(o2.cpu().numpy(), g['cutblur%d.2' % seed])
    for seed in range(3):
        np.random.seed(200 + seed)
        o1, o2 = augment.rgb(_t(g['a5']), _t(g['b5']), prob=1.0)
        assert np.array_equal(o1.cpu().numpy(), g['rgb%d.1' % seed]) and np.array_equal(o2.cpu().numpy(), g['rgb%d.2' % seed])
    for seed in range(6):
        np.random.seed(300 + seed)
        a, b = _t(g['a5']), _t(g['b5'])
        o1, o2 = augment.apply_augment(a, b, ['none', 'cutblur', 'rgb'], [1.0, 1.0, 1.0], [1.0, 0.7, 1.0], mix_p=[0.2, 0.5, 0.3])
        assert np.array_equal(o1.cpu().numpy(), g['mix%d.1' % seed]) and np.array_equal(o2.cpu().numpy(), g['mix%d.2' % seed])
        assert o1.data_ptr() != a.data_ptr() and o2.data_ptr() != b.data_ptr()       # fresh tensors, like the clones
    # blend: v * im + (1 - v) * colour[b, n, c]  (the colour comes from the device RNG; given explicitly here)
    a, b = _t(g['a5']), _t(g['b5'])
    col = torch.rand(a.shape[0], a.shape[1], 3, 1, 1, device=dev())
    o1, o2 = RF.augment_clips(a, b, v=0.7, colour=col)
    check('blend 1', o1, (0.7 * a + (1 - 0.7) * col).cpu(), 1e-6)
    check('blend 2', o2, (0.7 * b + (1 - 0.7) * col).cpu(), 1e-6)
    np.random.seed(7)
    torch.manual_seed(7)
    o1, o2 = augment.blend(a, b, prob=1.0, alpha=0.6)
    assert o1.shape == a.shape and float((o1 - a).abs().max()) > 0
    hi = torch.rand(2, 3, 3, 48, 64, device=dev())   # x4 pair: sizes differ, size-agnostic augmentations still work
    np.random.seed(8)
    p1, p2 = augment.rgb(hi, a, prob=1.0)
    np.random.seed(8)
    perm = list(np.random.permutation(3)) if np.random.rand(1) < 1.0 else None
    assert torch.equal(p1, hi[:, :, perm]) and torch.equal(p2, a[:, :, perm])


# ------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize('wd', [0.0, 1e-2])
def test_flat_adam_matches_torch_adam(wd):
    from realvsr_amd.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(16, 3, 3, 3), (16,), (5, 16, 1, 1), (5,), (7, 13)]
    ours = [torch.nn.Parameter(torch.randn(s, device=dev())) for s in shapes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    groups_o = [{'params': ours[:2], 'lr': 3e-3}, {'params': ours[2:]}]
    groups_t = [{'params': theirs[:2], 'lr': 3e-3}, {'params': theirs[2:]}]
    opt_o = FlatAdam(groups_o, lr=1e-3, betas=(0.9, 0.99), weight_decay=wd)
    opt_t = torch.optim.Adam(groups_t, lr=1e-3, betas=(0.9, 0.99), weight_decay=wd, foreach=False)
    for step in range(5):
        opt_o.zero_grad()
        opt_t.zero_grad()
        for po, pt in zip(ours, theirs):
            gr = torch.randn(po.shape, device=dev()) * (10.0 ** -step)
            po.grad = gr.clone()   # (after zero_grad() p.grad is None; step() copies a foreign gradient into the flat buffer)
            pt.grad = gr.clone()
        if step == 3:
            opt_o.param_groups[1]['lr'] = 5e-4    # schedulers edit param_groups in place (base_model.py:36-60)
            opt_t.param_groups[1]['lr'] = 5e-4
        opt_o.step()
        opt_t.step()
    for i, (po, pt) in enumerate(zip(ours, theirs)):
        check('param %d' % i, po, pt.detach().cpu(), 2e-6)
        check('exp_avg_sq %d' % i, opt_o.state[po]['exp_avg_sq'], opt_t.state[pt]['exp_avg_sq'].cpu(), 2e-6)
    # state_dict round trip keeps the moments inside the flat buffers
    sd = opt_o.state_dict()
    opt_o.load_state_dict(sd)
    assert opt_o.state[ours[0]]['exp_avg'].data_ptr() == opt_o.exp_avg.data_ptr() + 4 * opt_o.buffers.offset[ours[0]]
    assert int(opt_o.state[ours[0]]['step']) == 5
    # p.grad = None means "no gradient this step" (FlatBuffers.zero_grad leaves it so): step() rebinds it to its zeroed flat view;
    # a gradient that lives elsewhere is copied home; a PARAMETER moved out of the flat buffer is an error
    opt_o.zero_grad()
    ours[0].grad = None
    ours[1].grad = torch.ones_like(ours[1])
    opt_o.step()
    b = opt_o.buffers
    assert ours[0].grad.data_ptr() == b.grad.data_ptr() + 4 * b.offset[ours[0]] and float(ours[0].grad.abs().sum()) == 0
    assert ours[1].grad.data_ptr() == b.grad.data_ptr() + 4 * b.offset[ours[1]] and float(ours[1].grad.min()) == 1.0
    ours[0].data = ours[0].data.clone()
    with pytest.raises(RuntimeError):
        opt_o.step()


# ------------------------------------------------------------------------------------------ the training step
def _train_opt(tag):
    net = dict(which_model_G='EDVR', nf=16, nc=3, nframes=3, groups=4, front_RBs=1, back_RBs=1, center=None, predeblur=False,
               HR_in=False, w_TSA=True)
    return {'model': 'VideoSR_AllPair_YCbCr_Split', 'dist': False, 'gpu_ids': [0], 'is_train': True, 'scale': 4, 'augment': None,
            'network_G': net, 'path': {'pretrain_model_G': None, 'strict_load': True},
            'train': {'pixel_criterion_y': 'lappyr', 'pixel_weight_y': 1.0, 'pixel_criterion_c': 'gw', 'pixel_weight_c': 0.5,
                      'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-3, 'beta1': 0.9, 'beta2': 0.99}}


@pytest.mark.parametrize('tag', ['cb', 'ssim'])
def test_optimize_parameters_vs_reference_model(gemm_mode, tag):
    """3 steps of VideoSRModel.optimize_parameters vs the reference's own VideoSRModel (train_step.npz).
    'cb': every op pinned by the reference.  'ssim': the shipped 'lappyr' criterion; its SSIM term is UNPINNED."""
    from weights import fill_state_dict
    from oracle import edvr_oracle as O
    from realvsr_amd import loss as L
    from realvsr_amd.VideoSR_model import create_model
    g = load_golden('train_step')
    torch.cuda.set_device(0)
    model = create_model(_train_opt(tag))
    fill_state_dict(model.netG, 808, offset_std=0.02)          # in-place copy: parameters stay inside the flat buffer
    model.optimizer_G.buffers.check_bound()
    if tag == 'cb':
        model.cri_pix_y = L.LapPyrLoss(num_levels=3, lf_mode='cb', hf_mode='cb', reduction='mean')
    gt_c = torch.from_numpy(g['GT_center'])
    GT = torch.zeros(2, 3, 3, 96, 128)
    GT[:, 1] = gt_c
    data = {'LQs': torch.from_numpy(g['LQs']), 'GT': GT}
    ltol = 2e-5 if gemm_mode == 'f32' else 1e-4
    before = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
    for step in range(1, 4):
        model.feed_data(data)
        model.optimize_parameters(step)
        log = model.get_current_log()
        want = g[tag + '.logs'][step - 1]
        for name, w in zip(('l_pix_y', 'l_pix_c', 'l_pix'), want):
            assert abs(log[name] - w) <= ltol * abs(w), (step, name, log[name], w)
        if step == 1:
            gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.netG.parameters())).item()
            want_g = float(g[tag + '.gnorm1'])
            assert abs(gnorm - want_g) <= (1e-3 if gemm_mode == 'f32' else 5e-3) * want_g, (gnorm, want_g)
    # parameters after 3 Adam steps: compare the UPDATE (after - before); every entry moved by ~lr per step
    num = den = 0.0
    worst = 0.0
    for k, v in model.netG.state_dict().items():
        key = tag + '.after.' + k
        if key not in g:
            continue
        ref_after = torch.from_numpy(g[key]).double()
        d_ref = ref_after - before[k].cpu().double()
        d_got = v.detach().cpu().double() - before[k].cpu().double()
        num += float((d_got - d_ref).pow(2).sum())
        den += float(d_ref.pow(2).sum())
        worst = max(worst, float((v.detach().cpu().double() - ref_after).abs().max()))
    rel = (num / den) ** 0.5
    print('post-step parameter update: rel l2 err %.3e, worst abs param diff %.3e' % (rel, worst))
    # Adam's first steps are sign-like (m / sqrt(v) ~ +-1): a gradient entry whose sign differs in the last bits of a
    # near-zero value moves by 2 lr.  L2 over all entries bounds that; the bound is loose only in bf16x3 mode.
    assert rel <= (2e-2 if gemm_mode == 'f32' else 6e-2), rel
    assert worst <= 2.5 * 3 * 1e-3
    # PSNR-Y of the post-step output vs the reference's post-step output (north_star: within 1e-3 dB vs a GT)
    model.feed_data(data)
    model.test()
    fake_y = model.fake_H[:, 0:1].cpu()
    ref_y = torch.from_numpy(g[tag + '.fake_H_y'])
    p_build = O.psnr_y_uint8(fake_y, gt_c[:, 0:1])
    p_ref = O.psnr_y_uint8(ref_y, gt_c[:, 0:1])
    print('post-step PSNR-Y vs GT: build %.6f dB, reference %.6f dB' % (p_build, p_ref))
    assert abs(p_build - p_ref) <= 1e-3
    check_l2('post-step fake_H[:, 0]', fake_y, ref_y, 2e-4 if gemm_mode == 'f32' else 2e-3)


def test_combine_model_vs_reference_model(gemm_mode):
    """2 steps of VideoSRModel(split=False).optimize_parameters vs the reference's Combine model class
    (VideoSR_AllPair_model_YCbCr_Combine.py:187-221; fixture train_step_combine.npz from make_golden.main_train_step_combine):
    l_tot = 1.0 * Charbonnier on all three channels + 0.5 * PyramidLoss(3, 'lap', 'cb') as the edge term.  Every op pinned."""
    from weights import fill_state_dict
    from oracle import edvr_oracle as O
    from realvsr_amd.VideoSR_model import create_model
    g = load_golden('train_step_combine')
    torch.cuda.set_device(0)
    opt = _train_opt('cb')
    opt['model'] = 'VideoSR_AllPair_YCbCr_Combine'
    opt['train'] = {'pixel_criterion': 'cb', 'pixel_weight': 1.0, 'edge_criterion': 'pyr', 'edge_weight': 0.5,
                    'feature_criterion': None, 'feature_weight': 0, 'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-3,
                    'beta1': 0.9, 'beta2': 0.99}
    model = create_model(opt)
    assert model.split is False and model.cri_edg is not None
    fill_state_dict(model.netG, 909, offset_std=0.02)
    model.optimizer_G.buffers.check_bound()
    gt_c = torch.from_numpy(g['GT_center'])
    GT = torch.zeros(2, 3, 3, 64, 96)
    GT[:, 1] = gt_c
    data = {'LQs': torch.from_numpy(g['LQs']), 'GT': GT}
    ltol = 2e-5 if gemm_mode == 'f32' else 1e-4
    before = {k: v.detach().clone() for k, v in model.netG.state_dict().items()}
    for step in (1, 2):
        model.feed_data(data)
        model.optimize_parameters(step)
        log = model.get_current_log()
        for name, w in zip(('l_tot', 'l_edg'), g['logs'][step - 1]):
            assert abs(log[name] - w) <= ltol * abs(w), (step, name, log[name], w)
        if step == 1:
            gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.netG.parameters())).item()
            want_g = float(g['gnorm1'])
            assert abs(gnorm - want_g) <= (1e-3 if gemm_mode == 'f32' else 5e-3) * want_g, (gnorm, want_g)
    num = den = 0.0
    for k, v in model.netG.state_dict().items():
        d_ref = torch.from_numpy(g['after.' + k]).double() - before[k].cpu().double()
        d_got = v.detach().cpu().double() - before[k].cpu().double()
        num += float((d_got - d_ref).pow(2).sum())
        den += float(d_ref.pow(2).sum())
    rel = (num / den) ** 0.5
    print('Combine model, parameter update after 2 steps: rel l2 err %.3e' % rel)
    assert rel <= (2e-2 if gemm_mode == 'f32' else 6e-2), rel
    model.feed_data(data)
    model.test()
    fake, ref = model.fake_H.cpu(), torch.from_numpy(g['fake_H'])
    p_build, p_ref = O.psnr_y_uint8(fake[:, 0:1], gt_c[:, 0:1]), O.psnr_y_uint8(ref[:, 0:1], gt_c[:, 0:1])
    print('post-step PSNR-Y vs GT: build %.6f dB, reference %.6f dB' % (p_build, p_ref))
    assert abs(p_build - p_ref) <= 1e-3
    check_l2('post-step fake_H', fake, ref, 2e-4 if gemm_mode == 'f32' else 2e-3)


def test_model_step_is_sync_free_and_augments():
    """optimize_parameters(log=False) + the device augmentation path run end to end (ft_tsa_only groups, cutblur/rgb)."""
    from realvsr_amd.VideoSR_model import create_model
    opt = _train_opt('cb')
    opt['network_G'] = dict(opt['network_G'], which_model_G='EDVR_NoUp', nf=64, w_TSA=True, groups=8)
    opt['train'] = dict(opt['train'], pixel_criterion_y='cb', ft_tsa_only=2)
    opt['augment'] = {'augs': ['none', 'cutblur', 'rgb', 'blend'], 'probs': [1.0, 1.0, 1.0, 1.0], 'alphas': [1.0, 0.7, 1.0, 0.6],
                      'mix_p': [0.1, 0.4, 0.3, 0.2]}
    torch.manual_seed(0)
    np.random.seed(0)
    model = create_model(opt)
    assert len(model.optimizer_G.param_groups) == 2
    gen = torch.Generator().manual_seed(5)
    data = {'LQs': torch.rand(2, 3, 3, 16, 24, generator=gen), 'GT': torch.rand(2, 3, 3, 16, 24, generator=gen)}
    tsa0 = model.netG.tsa_fusion.tAtt_1.weight.detach().clone()
    first0 = model.netG.conv_first.weight.detach().clone()
    for step in range(1, 4):
        model.feed_data(data)
        model.optimize_parameters(step, log=False)
        if step == 1:   # ft_tsa_only: the non-TSA group has lr 0 (..._Split.py:159-161,164-165)
            assert torch.equal(model.netG.conv_first.weight.detach(), first0)
            assert not torch.equal(model.netG.tsa_fusion.tAtt_1.weight.detach(), tsa0)
    assert torch.isfinite(model.loss_terms['l_pix']).item() and model.get_current_log() == {}


def test_packed_weight_cache_is_bit_identical_and_batched():
    """functional.PackedWeights: bf16 hi/lo weight images are packed once per optimizer step (one batched launch
    after FlatAdam.step) instead of once per conv call.  Three training steps with the cache match three steps without it (first
    loss bit for bit, parameters within the run-to-run noise of the atomic scatter); edits that bypass the optimizer (in-place under no_grad: version bump; load_state_dict)
    are picked up; after the first step every conv / DCN forward of a step is a cache hit."""
    from weights import fill_state_dict
    from realvsr_amd import functional as RF
    from realvsr_amd.VideoSR_model import create_model
    torch.cuda.set_device(0)
    gen = torch.Generator().manual_seed(3)
    data = {'LQs': torch.rand(2, 3, 3, 24, 32, generator=gen), 'GT': torch.rand(2, 3, 3, 96, 128, generator=gen)}

    def run(enabled):
        RF.packed_weights.invalidate()
        RF.packed_weights.enabled = enabled
        RF.packed_weights.stats = {'hits': 0, 'packs': 0, 'batched': 0}
        torch.manual_seed(1)
        model = create_model(_train_opt('cb'))
        fill_state_dict(model.netG, 808, offset_std=0.02)
        losses = []
        for step in (1, 2, 3):
            model.feed_data(data)
            model.optimize_parameters(step)
            losses.append(model.get_current_log()['l_pix'])
            if step == 2:   # an edit behind the optimizer's back, through torch (version bump): must be seen by step 3
                with torch.no_grad():
                    model.netG.conv_first.weight.mul_(1.01)
                    model.netG.pcd_align.L1_dcnpack.weight.mul_(0.99)
        return model.optimizer_G.buffers.param.detach().clone(), losses, dict(RF.packed_weights.stats)

    try:
        p_on, l_on, st_on = run(True)
        p_off, l_off, st_off = run(False)
        p_off2, l_off2, _ = run(False)
        p_off3, _, _ = run(False)
    finally:
        RF.packed_weights.enabled = True
        RF.packed_weights.invalidate()
    # the forward pass has no atomics: the first loss is bit-identical; later steps carry the run-to-run noise of the float atomics
    # in the DCN grad_input flush (two runs WITHOUT the cache differ by it as well) -- the cache must not add to that noise floor
    assert l_on[0] == l_off[0] == l_off2[0]
    # (that noise is heavy-tailed -- a near-zero gradient entry that changes sign moves its parameter by 2 lr under Adam -- so the floor is the
    # largest of three off / off distances and the margin is wide; a STALE image shows in the losses below, by orders of magnitude)
    noise = max((a - b).double().norm().item() for a, b in ((p_off, p_off2), (p_off, p_off3), (p_off2, p_off3)))
    diff = (p_on - p_off).double().norm().item()
    print('parameters after 3 steps: cache on vs off %.3e, off vs off %.3e (of %.3e)' % (diff, noise, p_off.double().norm().item()))
    assert diff <= 10 * noise + 1e-5 * p_off.double().norm().item()
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(l_on, l_off))
    assert st_off == {'hits': 0, 'packs': 0, 'batched': 0}
    # one batched launch per optimizer step; individual packs only on first sight of an image (step 1) and for the two edited weights
    assert st_on['batched'] == 3 and st_on['hits'] >= 1.9 * (st_on['packs'] - 4) > 0
    print('packed-weight cache:', st_on)
