"""Hot-path losses with the reference's class names and constructor arguments
(codes/models/loss.py:10-23 CharbonnierLoss, :160-192 PyramidLoss, :195-224 LapPyrLoss).

``lf_mode='ssim'`` / ``hf_mode='ssim'`` call the third-party ``IQA_pytorch.SSIM`` in the
reference (loss.py:7,203,209) -- an unpinned pip dependency that is not vendored and is absent
here, so that mode is not implemented (parity would be unpinned); the in-tree 'cb' mode is."""
import torch
import torch.nn as nn

from . import functional as RF
from . import util


class CharbonnierLoss(nn.Module):
    """Charbonnier Loss (L1)"""

    def __init__(self, eps=1e-6, reduction='mean'):
        super(CharbonnierLoss, self).__init__()
        self.eps = eps
        self.reduction = reduction

    def forward(self, x, y):
        return RF.charbonnier(x, y, self.eps, 'mean' if self.reduction == 'mean' else 'sum')


class GWLoss(nn.Module):
    """Gradient Weighted Loss (codes/models/loss.py:54-80) -- the CbCr term of the reference's training loss
    (VideoSR_AllPair_model_YCbCr_Split.py:184).  The Sobel filters are constants of the HIP kernel."""

    def __init__(self, w=4, reduction='mean'):
        super(GWLoss, self).__init__()
        self.w = w
        self.reduction = reduction

    def forward(self, x1, x2):
        return RF.gw_loss(x1, x2, self.w, 'mean' if self.reduction == 'mean' else 'sum')


class PyramidLoss(nn.Module):
    """Pyramid Loss"""

    def __init__(self, num_levels=3, pyr_mode='gau', loss_mode='l1', reduction='mean'):
        super(PyramidLoss, self).__init__()
        self.num_levels = num_levels
        self.pyr_mode = pyr_mode
        self.loss_mode = loss_mode
        assert self.pyr_mode == 'gau' or self.pyr_mode == 'lap'
        if self.loss_mode == 'l1':
            self.loss = nn.L1Loss(reduction=reduction)
        elif self.loss_mode == 'l2':
            self.loss = nn.MSELoss(reduction=reduction)
        elif self.loss_mode == 'cb':
            self.loss = CharbonnierLoss(reduction=reduction)
        else:
            raise ValueError()

    def forward(self, x, y):
        pyr = util.gau_pyramid if self.pyr_mode == 'gau' else util.lap_pyramid
        pyr_x = pyr(img=x, max_levels=self.num_levels)
        pyr_y = pyr(img=y, max_levels=self.num_levels)
        loss = 0
        for i in range(self.num_levels):
            loss = loss + self.loss(pyr_x[i], pyr_y[i])
        return loss


class LapPyrLoss(nn.Module):
    """Pyramid Loss"""

    def __init__(self, num_levels=3, lf_mode='ssim', hf_mode='cb', reduction='mean'):
        super(LapPyrLoss, self).__init__()
        self.num_levels = num_levels
        self.lf_mode = lf_mode
        self.hf_mode = hf_mode
        for mode in (lf_mode, hf_mode):
            if mode == 'ssim':
                raise NotImplementedError("'ssim' needs the un-vendored IQA_pytorch package (parity unpinned); "
                                          "use lf_mode='cb'")
            if mode != 'cb':
                raise ValueError()
        self.lf_loss = CharbonnierLoss(reduction=reduction)
        self.hf_loss = CharbonnierLoss(reduction=reduction)

    def forward(self, x, y):
        pyr_x = util.laplacian_pyramid(img=x, max_levels=self.num_levels)
        pyr_y = util.laplacian_pyramid(img=y, max_levels=self.num_levels)
        loss = self.lf_loss(pyr_x[-1], pyr_y[-1])
        for i in range(self.num_levels - 1):
            loss = loss + self.hf_loss(pyr_x[i], pyr_y[i])
        return loss
