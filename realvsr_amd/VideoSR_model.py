"""The caller of the hot path: one training / test step with the reference's model API.

Mirrors ``VideoSRModel`` of codes/models/VideoSR_AllPair_model_YCbCr_Split.py (:22-151 constructor, :154-157 feed_data,
:163-191 optimize_parameters, :193-197 test) and of ..._YCbCr_Combine.py (:187-221), plus ``create_model``
(codes/models/__init__.py:5-17): same ``opt`` dictionary keys, same method names, same step sequence

    zero_grad -> (augment) -> netG(var_L) -> w_y * cri_y(fake[:, 0:1], GT[:, c, 0:1]) + w_c * cri_c(fake[:, 1:3], GT[:, c, 1:3])
              -> backward -> Adam step

MI355X wiring instead of DataParallel / DistributedDataParallel + torch.optim.Adam:
  * parameters, gradients and Adam moments live in flat buffers (optim.FlatAdam): zero_grad is one memset, the
    optimizer step one kernel;
  * with ``opt['dist']`` every rank (one process per GPU) starts from rank 0's parameters and the gradient buffer is
    all-reduced in buckets over RCCL while backward is still running (dist.BucketedGradAllReduce);
  * augmentation runs as one kernel on the device-resident clip pair (augment.apply_augment).
Out of scope: LR schedulers, logging, checkpoint cadence, the VGG feature loss and the GAN models.
"""
import os
from collections import OrderedDict

import torch

from . import VideoSR_archs as networks
from . import augment as augments
from . import loss as L
from .dist import BucketedGradAllReduce, broadcast_parameters
from .optim import FlatAdam


def _criterion(name, nc, role):
    """The loss table of the reference's constructor (..._Split.py:45-85, ..._Combine.py:44-82)."""
    if name == 'l1':
        return L.L1Loss(reduction='mean')
    if name == 'l2':
        return L.MSELoss(reduction='mean')
    if name == 'cb':
        return L.CharbonnierLoss(reduction='mean')
    if name == 'hb':
        return L.HuberLoss(reduction='mean')
    if name == 'gw' and role != 'combine':
        return L.GWLoss(w=4, reduction='mean')
    if name == 'pyr':
        return L.PyramidLoss(num_levels=3, pyr_mode='lap' if role == 'edge' else 'gau', loss_mode='cb', reduction='mean')
    if name == 'lappyr':
        return L.LapPyrLoss(num_levels=3, lf_mode='ssim', hf_mode='cb', reduction='mean')
    # 'msssim' (IQA_pytorch.MS_SSIM) is selected by no shipped option file and is not on the hot path
    raise NotImplementedError('Loss type [{:s}] is not recognized.'.format(str(name)))


class VideoSRModel:
    def __init__(self, opt, split=True):
        self.opt = opt
        self.split = split
        if opt.get('gpu_ids', 0) is None or not torch.cuda.is_available():
            raise NotImplementedError('realvsr_amd runs on MI355X only: there is no CPU path (gpu_ids=None)')
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.is_train = opt['is_train']
        self.schedulers, self.optimizers = [], []
        self.dist = bool(opt.get('dist'))
        self.rank = torch.distributed.get_rank() if self.dist else -1
        train_opt = opt.get('train') or {}

        self.netG = networks.define_G(opt).to(self.device)
        self.load()
        if self.dist:
            broadcast_parameters(self.netG)
        self.reducer = None
        if not self.is_train:
            return
        self.netG.train()
        nc = opt['network_G'].get('nc', 3)
        if split:
            self.cri_pix_y = _criterion(train_opt['pixel_criterion_y'], nc, 'y')
            self.l_pix_w_y = train_opt['pixel_weight_y']
            self.cri_pix_c = _criterion(train_opt['pixel_criterion_c'], nc, 'c')
            self.l_pix_w_c = train_opt['pixel_weight_c']
        else:
            self.cri_pix = _criterion(train_opt['pixel_criterion'], nc, 'combine')
            self.l_pix_w = train_opt['pixel_weight']
            self.cri_edg = None
            if train_opt.get('edge_criterion') and train_opt.get('edge_weight'):
                self.cri_edg = _criterion(train_opt['edge_criterion'], nc, 'edge')
                self.l_edg_w = train_opt['edge_weight']
            if train_opt.get('feature_criterion') and train_opt.get('feature_weight'):
                raise NotImplementedError('the VGG feature loss is outside the hot path (SURVEY.md section 2)')

        trainable = [(k, v) for k, v in self.netG.named_parameters() if v.requires_grad]
        if train_opt.get('ft_tsa_only'):
            groups = [{'params': [v for k, v in trainable if 'tsa_fusion' not in k], 'lr': train_opt['lr_G']},
                      {'params': [v for k, v in trainable if 'tsa_fusion' in k], 'lr': train_opt['lr_G']}]
        else:
            groups = [v for _, v in trainable]
        self.optimizer_G = FlatAdam(groups, lr=train_opt['lr_G'], weight_decay=train_opt.get('weight_decay_G') or 0,
                                    betas=(train_opt['beta1'], train_opt['beta2']))
        self.optimizers.append(self.optimizer_G)
        if self.dist:
            self.reducer = BucketedGradAllReduce(None, bucket_mb=float(os.environ.get('RVSR_BUCKET_MB', train_opt.get('bucket_mb', 4.0))),
                                                 buffers=self.optimizer_G.buffers, broadcast=False,
                                                 force=bool(train_opt.get('force_allreduce')))
        self.log_dict = OrderedDict()

    # ------------------------------------------------------------------ data
    def feed_data(self, data, need_GT=True):
        self.var_L = data['LQs'].to(self.device)
        if need_GT:
            self.var_H = data['GT'].to(self.device)

    # ------------------------------------------------------------------ one optimisation step
    def set_params_lr_zero(self):
        self.optimizers[0].param_groups[0]['lr'] = 0

    def forward_loss(self):
        """netG forward + the weighted criteria; returns (total, OrderedDict of the named terms) as device scalars."""
        self.fake_H = self.netG(self.var_L)
        center_idx = self.var_L.size(1) // 2
        gt = self.var_H[:, center_idx] if self.var_H.dim() == 5 else self.var_H   # [B, N, C, H, W] (or centre frame only)
        terms = OrderedDict()
        if self.split:
            terms['l_pix_y'] = self.l_pix_w_y * self.cri_pix_y(self.fake_H[:, 0:1], gt[:, 0:1])
            terms['l_pix_c'] = self.l_pix_w_c * self.cri_pix_c(self.fake_H[:, 1:3], gt[:, 1:3])
            total = terms['l_pix_y'] + terms['l_pix_c']
            terms['l_pix'] = total
        else:
            total = self.l_pix_w * self.cri_pix(self.fake_H, gt.contiguous())
            if self.cri_edg is not None:
                terms['l_edg'] = self.l_edg_w * self.cri_edg(self.fake_H, gt.contiguous())
                total = total + terms['l_edg']
            terms['l_tot'] = total
        return total, terms

    def optimize_parameters(self, step, log=True):
        """``log=False`` skips the .item() host syncs of the reference's log_dict (the values stay on the device in
        ``self.loss_terms``): the step is then free of host synchronisation."""
        train_opt = self.opt['train']
        if train_opt.get('ft_tsa_only') and step < train_opt['ft_tsa_only']:
            self.set_params_lr_zero()
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.optimizer_G.zero_grad()
        aug = self.opt.get('augment')
        if aug:
            self.var_H, self.var_L = augments.apply_augment(self.var_H, self.var_L, aug['augs'], aug['probs'],
                                                            aug['alphas'], aug['mix_p'])
        total, terms = self.forward_loss()
        total.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer_G.step()
        self.loss_terms = terms
        if log:
            for k, v in terms.items():
                self.log_dict[k] = v.item()

    def test(self):
        self.netG.eval()
        with torch.no_grad():
            self.fake_H = self.netG(self.var_L)
        self.netG.train()

    # ------------------------------------------------------------------ bookkeeping
    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_GT=True):
        out = OrderedDict()
        out['LQs'] = self.var_L.detach()[0].float().cpu()
        out['HQ'] = self.fake_H.detach()[0].float().cpu()
        if need_GT:
            out['GT'] = self.var_H.detach()[0].float().cpu()
        return out

    def get_current_learning_rate(self):
        return [g['lr'] for g in self.optimizers[0].param_groups]

    def load(self):
        path = (self.opt.get('path') or {}).get('pretrain_model_G')
        if path is not None:
            self.load_network(path, self.netG, self.opt['path'].get('strict_load', True))

    def load_network(self, load_path, network, strict=True):
        sd = torch.load(load_path, map_location='cpu')
        network.load_state_dict(OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items()),
                                strict=strict)

    def save_network(self, network, save_path):
        torch.save(OrderedDict((k, v.cpu()) for k, v in network.state_dict().items()), save_path)


def create_model(opt):
    model = opt['model']
    if model == 'VideoSR_AllPair_YCbCr_Split':
        return VideoSRModel(opt, split=True)
    if model == 'VideoSR_AllPair_YCbCr_Combine':
        return VideoSRModel(opt, split=False)
    # VideoSRGAN_AllPair_YCbCr_Split (discriminators, GAN losses) is outside the hot path
    raise NotImplementedError('Model [{:s}] not recognized.'.format(str(model)))
