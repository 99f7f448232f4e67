#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference Python (build container only).

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (read-only).  The reference Python never travels to the GPU box; only
the .npz data written here does.  Recipe = SURVEY.md Appendix B:
  * stub the third-party modules the reference imports but the hot path never calls,
  * replace the CUDA-only ``modulated_deform_conv`` with the CPU oracle (oracle/dcn_oracle.py),
  * drive the reference's own PCD_Align / TSA_Fusion / EDVR / EDVR_NoUp / pyramid / loss code
    with seeded tensors and record inputs, weights, outputs and gradients.
Every array is float32 (or float64 where noted); files are kept small (tiny nf / sizes).
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/codes'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
from weights import fill_state_dict  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)
    for name in ['kornia', 'cv2', 'ffmpeg', 'lmdb', 'IQA_pytorch', 'torchvision', 'torchvision.utils',
                 'models.archs.dcn.deform_conv_cuda']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    for cls in ['MS_SSIM', 'DISTS', 'LPIPSvgg']:
        setattr(sys.modules['IQA_pytorch'], cls, type(cls, (torch.nn.Module,), {}))
    # IQA_pytorch is third-party and absent: its SSIM is the restatement in oracle/ssim_oracle.py (PARITY UNPINNED);
    # it is only reached by the fixtures that say so (lf_mode='ssim')
    from oracle.ssim_oracle import SSIM as oracle_ssim
    sys.modules['IQA_pytorch'].SSIM = oracle_ssim
    sys.modules['torchvision.utils'].make_grid = lambda *a, **k: None
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    import models.archs.dcn.deform_conv  # noqa: F401
    from oracle.dcn_oracle import modulated_deform_conv as oracle_dcn
    sys.modules['models.archs.dcn.deform_conv'].modulated_deform_conv = oracle_dcn
    import models.archs.EDVR_arch as EDVR_arch
    import models.loss as loss
    import utils.util as util
    return EDVR_arch, loss, util, sys.modules['models.archs.dcn.deform_conv']


def np_sd(module):
    return {'sd.' + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def np_grads(module):
    return {'grad.' + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def randomize_offset_convs(module, std):
    """conv_offset_mask is zero-initialised in the reference (deform_conv.py:270-272); give it
    non-trivial weights so that offsets/masks are exercised."""
    g = torch.Generator().manual_seed(99)
    for name, p in module.named_parameters():
        if 'conv_offset_mask' in name:
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * std)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def main():
    torch.set_num_threads(4)
    EDVR_arch, loss_mod, util, dc = import_reference()

    # ---- 1. DCN pack: reference ModulatedDeformConvPack wiring + oracle operator ---------------
    torch.manual_seed(1)
    pack = dc.ModulatedDeformConvPack(16, 12, 3, stride=1, padding=1, dilation=1,
                                      deformable_groups=4, extra_offset_mask=True)
    randomize_offset_convs(pack, 0.3)  # |offset| up to several px -> border / out-of-range taps
    x = torch.randn(2, 16, 7, 9, requires_grad=True)
    feat = torch.randn(2, 16, 7, 9, requires_grad=True)
    out = pack([x, feat])
    gout = torch.randn_like(out)
    out.backward(gout)
    save('dcn_pack', x=x.detach().numpy(), feat=feat.detach().numpy(), out=out.detach().numpy(),
         gout=gout.numpy(), gx=x.grad.numpy(), gfeat=feat.grad.numpy(), **np_sd(pack), **np_grads(pack))

    # ---- 2. raw operator with hand-made offsets (border, integer, far out of range) ------------
    torch.manual_seed(2)
    B, C, Co, dg, H, W = 2, 8, 6, 2, 6, 10
    x = torch.randn(B, C, H, W, requires_grad=True)
    offset = (torch.randn(B, dg * 18, H, W) * 2.0)
    offset[0, :, 0, :] = -1.5          # samples in (-1, 0) and below -1
    offset[0, :, -1, :] = 1.25         # samples in (H-1, H)
    offset[1, :, :, 0] = -40.0         # far out of range
    offset[1, :, 2, :] = 1.0           # exact integer offsets
    offset.requires_grad_(True)
    mask = torch.rand(B, dg * 9, H, W, requires_grad=True)
    weight = torch.randn(Co, C, 3, 3, requires_grad=True)
    bias = torch.randn(Co, requires_grad=True)
    out = dc.modulated_deform_conv(x, offset, mask, weight, bias, 1, 1, 1, 1, dg)
    gout = torch.randn_like(out)
    out.backward(gout)
    save('dcn_op', x=x.detach().numpy(), offset=offset.detach().numpy(), mask=mask.detach().numpy(),
         weight=weight.detach().numpy(), bias=bias.detach().numpy(), out=out.detach().numpy(),
         gout=gout.numpy(), gx=x.grad.numpy(), goffset=offset.grad.numpy(), gmask=mask.grad.numpy(),
         gweight=weight.grad.numpy(), gbias=bias.grad.numpy(), dg=np.int32(dg))

    # ---- 3. PCD_Align ---------------------------------------------------------------------------
    torch.manual_seed(3)
    nf, groups = 16, 4
    pcd = EDVR_arch.PCD_Align(nf=nf, groups=groups)
    randomize_offset_convs(pcd, 0.05)
    nbr = [torch.randn(1, nf, 16 >> l, 24 >> l, requires_grad=True) for l in range(3)]
    ref = [torch.randn(1, nf, 16 >> l, 24 >> l, requires_grad=True) for l in range(3)]
    out = pcd(nbr, ref)
    gout = torch.randn_like(out)
    out.backward(gout)
    arrs = {}
    for l in range(3):
        arrs['nbr%d' % l] = nbr[l].detach().numpy()
        arrs['ref%d' % l] = ref[l].detach().numpy()
        arrs['gnbr%d' % l] = nbr[l].grad.numpy()
        arrs['gref%d' % l] = ref[l].grad.numpy()
    save('pcd_align', out=out.detach().numpy(), gout=gout.numpy(), nf=np.int32(nf), groups=np.int32(groups),
         **arrs, **np_sd(pcd), **np_grads(pcd))

    # ---- 4. TSA_Fusion --------------------------------------------------------------------------
    torch.manual_seed(4)
    tsa = EDVR_arch.TSA_Fusion(nf=16, nframes=3, center=1)
    al = torch.randn(2, 3, 16, 12, 20, requires_grad=True)
    out = tsa(al)
    gout = torch.randn_like(out)
    out.backward(gout)
    save('tsa_fusion', aligned=al.detach().numpy(), out=out.detach().numpy(), gout=gout.numpy(),
         galigned=al.grad.numpy(), **np_sd(tsa), **np_grads(tsa))

    # ---- 5. EDVR (x4, TSA) and EDVR_NoUp (no TSA), tiny ----------------------------------------
    for name, cls, kw in [('edvr_tsa', EDVR_arch.EDVR, dict(nf=16, nframes=3, groups=4, front_RBs=2, back_RBs=2, w_TSA=True)),
                          ('edvr_noup', EDVR_arch.EDVR_NoUp, dict(nf=64, nframes=3, groups=8, front_RBs=1, back_RBs=1, w_TSA=False))]:
        torch.manual_seed(5)
        net = cls(nc=3, center=None, predeblur=False, HR_in=False, **kw)
        if name == 'edvr_noup':  # nf must be 64 (EDVR_arch.py:352): weights re-created from a seed
            fill_state_dict(net, 77)
        else:
            randomize_offset_convs(net, 0.02)
        gen = torch.Generator().manual_seed(1234)
        hw = (16, 24) if name == 'edvr_tsa' else (8, 12)
        x = torch.rand(1, kw['nframes'], 3, *hw, generator=gen)
        out = net(x)
        gt = torch.rand(out.shape, generator=torch.Generator().manual_seed(1235))
        crit = loss_mod.LapPyrLoss(num_levels=3, lf_mode='cb', hf_mode='cb', reduction='mean')
        l = crit(out[:, 0:1], gt[:, 0:1]) + loss_mod.CharbonnierLoss()(out[:, 1:3], gt[:, 1:3])
        l.backward()
        gnorm = torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters()))
        sd = np_sd(net)
        grads = np_grads(net)
        if name == 'edvr_noup':  # nf=64 model: keep the file small, store only a few gradients
            sd = {}
            grads = {k: v for k, v in grads.items() if k.split('grad.')[1] in (
                'conv_first.weight', 'conv_first.bias', 'pcd_align.L1_dcnpack.weight',
                'pcd_align.cas_dcnpack.conv_offset_mask.bias', 'conv_last.weight')}
        save(name, x=x.numpy(), out=out.detach().numpy(), gt=gt.numpy(), loss=np.float64(l.item()),
             gnorm=np.float64(gnorm.item()), **{k: np.asarray(v) for k, v in kw.items()}, **sd, **grads)

    # ---- 6. pyramids on integer-valued inputs (exactly representable => bit-exact indexing) -----
    arrs = {}
    for tag, (C, H, W) in {'a': (1, 64, 64), 'b': (3, 36, 52), 'c': (1, 8, 12)}.items():
        g = torch.Generator().manual_seed(60 + C + H)
        img = torch.randint(0, 16, (2, C, H, W), generator=g).float()
        k = util.gauss_kernel(channels=C)
        arrs['img_' + tag] = img.numpy()
        for i, lv in enumerate(util.laplacian_pyramid(img, k, 3)):
            arrs['laplacian_%s_%d' % (tag, i)] = lv.numpy()
        for i, lv in enumerate(util.lap_pyramid(img, k, 2)):
            arrs['lap_%s_%d' % (tag, i)] = lv.numpy()
        for i, lv in enumerate(util.gau_pyramid(img, k, 3)):
            arrs['gau_%s_%d' % (tag, i)] = lv.numpy()
    save('pyramid_int', **arrs)

    # ---- 7. losses + gradients on float inputs --------------------------------------------------
    torch.manual_seed(7)
    arrs = {}
    for tag, C in (('y', 1), ('rgb', 3)):
        x = torch.rand(2, C, 24, 40, requires_grad=True)
        y = torch.rand(2, C, 24, 40)
        arrs['x_' + tag], arrs['y_' + tag] = x.detach().numpy(), y.numpy()
        for lname, crit in [('lappyr_cb', loss_mod.LapPyrLoss(3, 'cb', 'cb', 'mean')),
                            ('lappyr_cb_sum', loss_mod.LapPyrLoss(2, 'cb', 'cb', 'sum')),
                            ('pyr_gau_cb', loss_mod.PyramidLoss(3, 'gau', 'cb', 'mean')),
                            ('pyr_lap_l1', loss_mod.PyramidLoss(2, 'lap', 'l1', 'mean')),
                            ('pyr_gau_l2', loss_mod.PyramidLoss(3, 'gau', 'l2', 'mean')),
                            ('cb', loss_mod.CharbonnierLoss()), ('gw', loss_mod.GWLoss(w=4)),
                            ('gw_sum', loss_mod.GWLoss(w=2, reduction='sum'))]:
            x.grad = None
            l = crit(x, y)
            l.backward()
            arrs['%s_%s' % (lname, tag)] = np.float64(l.item())
            arrs['g_%s_%s' % (lname, tag)] = x.grad.numpy().copy()
    save('losses', **arrs)


def main_infer():
    """infer.npz: the steps either side of the path at test time:
    index_generation tables, flip x4 ensemble of a tiny EDVR_NoUp, YCbCr -> BGR uint8 quantisation
    (test_RealVSR_wi_GT.py:114-123)."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    import data.util as data_util
    arrs = {}
    # ---- index_generation (data/util.py:169-214): every centre for a few (max_n, N), all four modes
    rows = []
    modes = ['replicate', 'reflection', 'new_info', 'circle']
    for mi, mode in enumerate(modes):
        for max_n in (5, 7, 12, 30):
            for N in (3, 5, 7):
                if N > max_n:
                    continue
                for crt in range(max_n):
                    rows.append([mi, max_n, N, crt] + data_util.index_generation(crt, max_n, N, padding=mode) + [-1] * (7 - N))
    arrs['index_table'] = np.array(rows, dtype=np.int32)
    # ---- colour conversion + quantisation exactly as the test script does it
    rng = np.random.RandomState(5)
    ycc = (rng.rand(3, 24, 40).astype(np.float32) * 1.3 - 0.15)       # values outside [0, 1] on purpose
    ycc[:, :2, :] = np.array([16, 128, 128], np.float32)[:, None, None] / 255.  # black
    ycc[:, 2:4, :] = np.array([235, 128, 128], np.float32)[:, None, None] / 255.  # white
    t = torch.from_numpy(ycc.copy())
    out = util.tensor2img(t, out_type=np.float32, reverse_channel=False)
    img = (np.clip(data_util.ycbcr2bgr(out), 0, 1) * 255.).round().astype(np.uint8)
    arrs['ycc'] = ycc
    arrs['bgr_u8'] = img
    # ---- flip x4 self-ensemble through the reference's own helper on a tiny EDVR_NoUp
    torch.manual_seed(3)
    net = EDVR_arch.EDVR_NoUp(nf=64, nc=3, nframes=3, groups=8, front_RBs=1, back_RBs=1, center=None, predeblur=False,
                              HR_in=False, w_TSA=True)
    fill_state_dict(net, 123)
    net.eval()
    x = torch.rand(1, 3, 3, 16, 24)
    arrs['flip_x'] = x.numpy()
    arrs['flip_single'] = util.single_forward(net, x).numpy()
    arrs['flip_x4'] = util.flipx4_forward(net, x).numpy()
    save('infer', **arrs)


def main_tdan():
    """tdan.npz: the reference's TDAN (models/archs/TDAN_arch.py) with the oracle DCN: output and a few
    gradients for two small configurations (scale 1 as RealVSR uses it, and scale 2 for the Upsampler path)."""
    import_reference()
    import models.archs.TDAN_arch as TDAN_arch
    arrs = {}
    for tag, scale, nframes, (H, W) in (('s1', 1, 3, (12, 20)), ('s2', 2, 3, (8, 12))):
        torch.manual_seed(11)
        net = TDAN_arch.TDAN(channel=3, nframes=nframes, scale=scale, nf=64, nb_f=1, nb_b=1, groups=8)
        fill_state_dict(net, 31, offset_std=0.05)
        x = torch.rand(1, nframes, 3, H, W, generator=torch.Generator().manual_seed(12))
        out = net(x)
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(13))
        out.backward(gout)
        arrs[tag + '.x'] = x.numpy()
        arrs[tag + '.out'] = out.detach().numpy()
        arrs[tag + '.gout'] = gout.numpy()
        for k in ('align.deform_conv_1.weight', 'align.deform_conv.conv_offset_mask.weight', 'align.bottle_neck.weight',
                  'align.initial_conv.bias', 'trunk.feature_extractor.0.weight', 'trunk.upsampler.1.weight'):
            arrs[tag + '.grad.' + k] = dict(net.named_parameters())[k].grad.numpy().copy()
        arrs[tag + '.keys'] = np.array(sorted(net.state_dict().keys()))
    save('tdan', **arrs)


def main_augment():
    """augment.npz: the reference's cutblur / rgb / apply_augment (data/augments_video_allpair.py) under fixed numpy
    seeds; 4-D images for cutblur (the shape its h/w indexing is written for) and 5-D clips for rgb."""
    import_reference()
    import data.augments_video_allpair as aug
    arrs = {}
    rs = np.random.RandomState(21)
    a4, b4 = rs.rand(2, 3, 20, 28).astype(np.float32), rs.rand(2, 3, 20, 28).astype(np.float32)
    a5, b5 = rs.rand(2, 3, 3, 12, 16).astype(np.float32), rs.rand(2, 3, 3, 12, 16).astype(np.float32)
    arrs.update(a4=a4, b4=b4, a5=a5, b5=b5)
    for seed in range(6):
        np.random.seed(100 + seed)
        o1, o2 = aug.cutblur(torch.from_numpy(a4.copy()), torch.from_numpy(b4.copy()), prob=1.0, alpha=0.7)
        arrs['cutblur%d.1' % seed], arrs['cutblur%d.2' % seed] = o1.numpy(), o2.numpy()
    for seed in range(3):
        np.random.seed(200 + seed)
        o1, o2 = aug.rgb(torch.from_numpy(a5.copy()), torch.from_numpy(b5.copy()), prob=1.0)
        arrs['rgb%d.1' % seed], arrs['rgb%d.2' % seed] = o1.numpy().copy(), o2.numpy().copy()
    for seed in range(6):
        np.random.seed(300 + seed)
        o1, o2 = aug.apply_augment(torch.from_numpy(a5.copy()), torch.from_numpy(b5.copy()), ['none', 'cutblur', 'rgb'],
                                   [1.0, 1.0, 1.0], [1.0, 0.7, 1.0], mix_p=[0.2, 0.5, 0.3])
        arrs['mix%d.1' % seed], arrs['mix%d.2' % seed] = o1.numpy().copy(), o2.numpy().copy()
    save('augment', **arrs)


def main_predeblur():
    """edvr_predeblur.npz: reference EDVR with predeblur=True (LR input), with HR_in=True, and with both: output and a
    few gradients, seeded weights (weights.fill_state_dict)."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    arrs = {}
    for tag, kw, (H, W) in (('pre', dict(predeblur=True, HR_in=False), (16, 24)), ('hr', dict(predeblur=False, HR_in=True), (32, 48)),
                            ('prehr', dict(predeblur=True, HR_in=True), (32, 32))):
        torch.manual_seed(21)
        net = EDVR_arch.EDVR(nf=64, nc=3, nframes=3, groups=8, front_RBs=1, back_RBs=1, center=None, w_TSA=False, **kw)
        fill_state_dict(net, 41, offset_std=0.05)
        x = torch.rand(1, 3, 3, H, W, generator=torch.Generator().manual_seed(22))
        out = net(x)
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(23))
        out.backward(gout)
        arrs[tag + '.x'], arrs[tag + '.out'], arrs[tag + '.gout'] = x.numpy(), out.detach().numpy(), gout.numpy()
        params = dict(net.named_parameters())
        for k in [n for n in params if n.startswith(('pre_deblur.RB_L3_1.conv1.w', 'pre_deblur.deblur_L2_conv.w', 'conv_1x1.w',
                                                     'conv_first_2.w', 'pre_deblur.conv_first_3.b', 'conv_last.b'))]:
            arrs[tag + '.grad.' + k] = params[k].grad.numpy().copy()
        arrs[tag + '.keys'] = np.array(sorted(net.state_dict().keys()))
    save('edvr_predeblur', **arrs)


def main_train_step():
    """train_step.npz: the reference's OWN VideoSRModel (codes/models/VideoSR_AllPair_model_YCbCr_Split.py) built through
    create_model(opt) on CPU (gpu_ids None), driven for 3 optimize_parameters() steps on one seeded batch: per-step
    loss terms, gradient norm of the first step, and the parameters after the last step.
      tag 'cb'   : cri_pix_y replaced by the reference's LapPyrLoss(3, 'cb', 'cb') -- every op in-tree => PINNED
      tag 'ssim' : the model's own 'lappyr' criterion, LapPyrLoss(3, 'ssim', 'cb'), with IQA_pytorch.SSIM = the
                   restatement in oracle/ssim_oracle.py => composition pinned, SSIM term UNPINNED
    Both: cri_pix_c = GWLoss(w=4), pixel_weight_y 1.0, pixel_weight_c 0.5, Adam lr 1e-3 betas (0.9, 0.99)."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    import models
    net_opt = dict(which_model_G='EDVR', nf=16, nc=3, nframes=3, groups=4, front_RBs=1, back_RBs=1, center=None,
                   predeblur=False, HR_in=False, w_TSA=True)
    arrs = {}
    for tag in ('cb', 'ssim'):
        opt = {'model': 'VideoSR_AllPair_YCbCr_Split', 'dist': False, 'gpu_ids': None, 'is_train': True, 'scale': 4,
               'augment': None, 'network_G': dict(net_opt), 'path': {'pretrain_model_G': None, 'strict_load': True},
               'train': {'pixel_criterion_y': 'lappyr', 'pixel_weight_y': 1.0, 'pixel_criterion_c': 'gw',
                         'pixel_weight_c': 0.5, 'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-3, 'beta1': 0.9,
                         'beta2': 0.99, 'lr_scheme': 'MultiStepLR_Restart', 'lr_steps': [1000], 'restarts': None,
                         'restart_weights': None, 'lr_gamma': 0.5, 'clear_state': None}}
        torch.manual_seed(8)
        model = models.create_model(opt)
        net = model.netG.module if hasattr(model.netG, 'module') else model.netG
        fill_state_dict(net, 808, offset_std=0.02)     # weights re-created from the seed by the tests (not stored)
        if tag == 'cb':
            model.cri_pix_y = loss_mod.LapPyrLoss(num_levels=3, lf_mode='cb', hf_mode='cb', reduction='mean')
        gen = torch.Generator().manual_seed(81)
        # 24x32 LR -> 96x128 HR: the low-pass band of the 3-level pyramid is 24x32 (>= the 11x11 SSIM window).
        # Only the centre GT frame is read by the model (:178-184); the others are zeros and are not stored.
        gt_c = torch.rand(2, 3, 96, 128, generator=gen)
        GT = torch.zeros(2, 3, 3, 96, 128)
        GT[:, 1] = gt_c
        data = {'LQs': torch.rand(2, 3, 3, 24, 32, generator=gen), 'GT': GT}
        if tag == 'cb':
            arrs['LQs'], arrs['GT_center'] = data['LQs'].numpy(), gt_c.numpy()
        logs = []
        for step in range(1, 4):
            model.feed_data(data)
            model.optimize_parameters(step)
            log = model.get_current_log()
            logs.append([log['l_pix_y'], log['l_pix_c'], log['l_pix']])
            if step == 1:
                arrs[tag + '.gnorm1'] = np.float64(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())).item())
        arrs[tag + '.logs'] = np.array(logs, dtype=np.float64)
        for k, v in net.state_dict().items():
            if tag == 'cb' or k.startswith(('conv_first', 'pcd_align.L1_dcnpack', 'tsa_fusion.sAtt_5', 'conv_last', 'upconv2.bias')):
                arrs[tag + '.after.' + k] = v.detach().numpy().copy()
        model.feed_data(data)
        model.test()
        arrs[tag + '.fake_H_y'] = model.fake_H[:, 0:1].numpy().copy()
    save('train_step', **arrs)


def main_train_step_combine():
    """train_step_combine.npz: the reference's Combine model class (codes/models/VideoSR_AllPair_model_YCbCr_Combine.py:187-221)
    built through create_model(opt) on CPU and driven for 2 optimize_parameters() steps: l_tot = 1.0 * CharbonnierLoss on all
    three channels + 0.5 * PyramidLoss(3, 'lap', 'cb') ('edge_criterion: pyr'); per-step l_tot / l_edg, gradient norm of the
    first step, parameters after the second step, post-step output.  Every op is in-tree => PINNED."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    import models
    net_opt = dict(which_model_G='EDVR', nf=16, nc=3, nframes=3, groups=4, front_RBs=1, back_RBs=1, center=None,
                   predeblur=False, HR_in=False, w_TSA=True)
    opt = {'model': 'VideoSR_AllPair_YCbCr_Combine', 'dist': False, 'gpu_ids': None, 'is_train': True, 'scale': 4,
           'augment': None, 'network_G': dict(net_opt), 'path': {'pretrain_model_G': None, 'strict_load': True},
           'train': {'pixel_criterion': 'cb', 'pixel_weight': 1.0, 'edge_criterion': 'pyr', 'edge_weight': 0.5,
                     'feature_criterion': None, 'feature_weight': 0, 'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-3,
                     'beta1': 0.9, 'beta2': 0.99, 'lr_scheme': 'MultiStepLR_Restart', 'lr_steps': [1000], 'restarts': None,
                     'restart_weights': None, 'lr_gamma': 0.5, 'clear_state': None}}
    torch.manual_seed(9)
    model = models.create_model(opt)
    net = model.netG.module if hasattr(model.netG, 'module') else model.netG
    fill_state_dict(net, 909, offset_std=0.02)     # weights re-created from the seed by the tests (not stored)
    gen = torch.Generator().manual_seed(91)
    gt_c = torch.rand(2, 3, 64, 96, generator=gen)
    GT = torch.zeros(2, 3, 3, 64, 96)
    GT[:, 1] = gt_c
    data = {'LQs': torch.rand(2, 3, 3, 16, 24, generator=gen), 'GT': GT}
    arrs = {'LQs': data['LQs'].numpy(), 'GT_center': gt_c.numpy()}
    logs = []
    for step in range(1, 3):
        model.feed_data(data)
        model.optimize_parameters(step)
        log = model.get_current_log()
        logs.append([log['l_tot'], log['l_edg']])
        if step == 1:
            arrs['gnorm1'] = np.float64(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters())).item())
    arrs['logs'] = np.array(logs, dtype=np.float64)
    for k, v in net.state_dict().items():
        arrs['after.' + k] = v.detach().numpy().copy()
    model.feed_data(data)
    model.test()
    arrs['fake_H'] = model.fake_H.numpy().copy()
    save('train_step_combine', **arrs)


def main_config3():
    """edvr_c3.npz: the architecture of BASELINE configs 3-5 (nf128, 7 frames, TSA, x4) through the reference's EDVR at
    32x48 (back_RBs reduced to 2 to keep the CPU run short): output, loss, gradient norm and a few gradients.  Weights
    from the seeded fill (too large to commit)."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    torch.manual_seed(5)
    kw = dict(nf=128, nframes=7, groups=8, front_RBs=5, back_RBs=2, w_TSA=True)
    net = EDVR_arch.EDVR(nc=3, center=None, predeblur=False, HR_in=False, **kw)
    fill_state_dict(net, 303, offset_std=0.03)
    x = torch.rand(1, 7, 3, 32, 48, generator=torch.Generator().manual_seed(1234))
    out = net(x)
    gt = torch.rand(out.shape, generator=torch.Generator().manual_seed(1235))
    l = loss_mod.LapPyrLoss(3, 'cb', 'cb', 'mean')(out[:, 0:1], gt[:, 0:1]) + loss_mod.GWLoss(w=4)(out[:, 1:3], gt[:, 1:3])
    l.backward()
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters()))
    params = dict(net.named_parameters())
    keep = ['conv_first.weight', 'feature_extraction.4.conv2.weight', 'fea_L3_conv1.weight', 'pcd_align.L3_dcnpack.weight',
            'pcd_align.L1_dcnpack.weight', 'pcd_align.L1_dcnpack.conv_offset_mask.bias', 'pcd_align.cas_dcnpack.bias',
            'pcd_align.L2_offset_conv2.bias', 'tsa_fusion.tAtt_2.bias', 'tsa_fusion.sAtt_L2.bias', 'recon_trunk.0.conv1.bias',
            'upconv2.bias', 'conv_last.weight']
    grads = {'grad.' + k: params[k].grad.numpy().copy() for k in keep}
    grads['grad.conv_first.weight'] = grads['grad.conv_first.weight'].copy()
    # large weight gradients: keep a slice (first 8 output channels) to bound the file size
    for k in list(grads):
        if grads[k].ndim == 4 and grads[k].size > 40000:
            grads[k] = grads[k][:8].copy()
    save('edvr_c3', x=x.numpy(), out=out.detach().numpy(), gt=gt.numpy(), loss=np.float64(l.item()),
         gnorm=np.float64(gnorm.item()), **{k: np.asarray(v) for k, v in kw.items()}, **grads)


def main_losses2():
    """losses2.npz: the remaining criteria of loss.py (HuberLoss, PyramidLoss 'hb') and the stand-alone pyramid helpers
    conv_gauss / upsample (utils/util.py:503-516), values + gradients.  Plus LapPyrLoss(3,'ssim','cb') with the
    restated SSIM (UNPINNED third-party term; the rest of the composition is the reference's)."""
    EDVR_arch, loss_mod, util, dc = import_reference()
    torch.manual_seed(17)
    arrs = {}
    x = torch.rand(2, 1, 48, 64, requires_grad=True)
    y = torch.rand(2, 1, 48, 64)
    x.data[0, 0, :4] = y[0, 0, :4] + 0.003     # inside Huber's quadratic zone (delta 0.01)
    arrs['x'], arrs['y'] = x.detach().numpy().copy(), y.numpy()
    for lname, crit in [('hb', loss_mod.HuberLoss()), ('hb_sum', loss_mod.HuberLoss(delta=0.05, reduction='sum')),
                        ('pyr_gau_hb', loss_mod.PyramidLoss(3, 'gau', 'hb', 'mean')),
                        ('pyr_lap_hb', loss_mod.PyramidLoss(2, 'lap', 'hb', 'mean')),
                        ('lappyr_ssim_UNPINNED', loss_mod.LapPyrLoss(3, 'ssim', 'cb', 'mean'))]:
        x.grad = None
        l = crit(x, y)
        l.backward()
        arrs[lname] = np.float64(l.item())
        arrs['g_' + lname] = x.grad.numpy().copy()
    for tag, C, H, W in (('a', 3, 20, 28), ('b', 1, 6, 8)):
        img = torch.rand(2, C, H, W, requires_grad=True)
        k = util.gauss_kernel(channels=C)
        for fname, fn in (('conv_gauss', lambda t: util.conv_gauss(t, k)), ('conv_gauss4', lambda t: util.conv_gauss(t, 4 * k)),
                          ('upsample', util.upsample)):
            img.grad = None
            out = fn(img)
            gout = torch.randn(out.shape)
            out.backward(gout)
            arrs['%s_%s.in' % (fname, tag)] = img.detach().numpy().copy()
            arrs['%s_%s.out' % (fname, tag)] = out.detach().numpy().copy()
            arrs['%s_%s.gout' % (fname, tag)] = gout.numpy()
            arrs['%s_%s.gin' % (fname, tag)] = img.grad.numpy().copy()
        ii = torch.randint(0, 16, (1, C, H, W)).float()
        arrs['int_%s.in' % tag] = ii.numpy()
        arrs['int_%s.conv_gauss' % tag] = util.conv_gauss(ii, k).numpy()
        arrs['int_%s.upsample' % tag] = util.upsample(ii).numpy()
    save('losses2', **arrs)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'train_step'):
        main_train_step()
    if which in ('all', 'train_step_combine'):
        main_train_step_combine()
    if which in ('all', 'config3'):
        main_config3()
    if which in ('all', 'losses2'):
        main_losses2()
    if which in ('all', 'predeblur'):
        main_predeblur()
    if which in ('all', 'augment'):
        main_augment()
    if which in ('all', 'main'):
        main()
    if which in ('all', 'infer'):
        main_infer()
    if which in ('all', 'tdan'):
        main_tdan()
