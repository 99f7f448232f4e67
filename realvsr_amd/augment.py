"""Training-time augmentation immediately before the hot path: mirrors
codes/data/augments_video_allpair.py (apply_augment :6-35, blend :38-51, cutblur :54-77, rgb :80-88) with the same
names, arguments, host-RNG call order (numpy global state) and in-place behaviour, so a seeded run selects the same
boxes / permutations as the reference.  These are rectangle copies and channel permutations on tensors that are
already on the GPU -- pure data movement on whatever device the inputs live on, no kernels of their own.
"""
import numpy as np
import torch


def apply_augment(im1, im2, augs, probs, alphas, mix_p=None):
    idx = np.random.choice(len(augs), p=mix_p)
    aug, prob, alpha = augs[idx], float(probs[idx]), float(alphas[idx])
    if aug == 'none':
        return im1.clone(), im2.clone()
    if aug == 'blend':
        return blend(im1.clone(), im2.clone(), prob=prob, alpha=alpha)
    if aug == 'cutblur':
        return cutblur(im1.clone(), im2.clone(), prob=prob, alpha=alpha)
    if aug == 'rgb':
        return rgb(im1.clone(), im2.clone(), prob=prob)
    raise ValueError('{} is not invalid.'.format(aug))


def blend(im1, im2, prob=1.0, alpha=0.6):
    """Blend both clips ([B, N, 3, H, W]) with one random colour per frame (torch RNG), weight v ~ U(alpha, 1)."""
    if alpha <= 0 or np.random.rand(1) >= prob:
        return im1, im2
    c = torch.empty((im2.size(0), im2.size(1), 3, 1, 1), device=im2.device).uniform_(0, 1)
    v = np.random.uniform(alpha, 1)
    return v * im1 + (1 - v) * c, v * im2 + (1 - v) * c   # broadcasting instead of the reference's repeat()


def cutblur(im1, im2, prob=1.0, alpha=1.0):
    """Paste a random box of im1 into im2 (in place), or of im2 into a copy of im1 (which becomes im2)."""
    if im1.size() != im2.size():
        raise ValueError('im1 and im2 have to be the same resolution.')
    if alpha <= 0 or np.random.rand(1) >= prob:
        return im1, im2
    cut_ratio = np.random.randn() * 0.01 + alpha
    h, w = im2.size(2), im2.size(3)          # (sic) the reference indexes dims 2, 3 of a 5-D clip as well
    ch, cw = int(h * cut_ratio), int(w * cut_ratio)
    cy = np.random.randint(0, h - ch + 1)
    cx = np.random.randint(0, w - cw + 1)
    if np.random.random() > 0.5:
        im2[..., cy:cy + ch, cx:cx + cw] = im1[..., cy:cy + ch, cx:cx + cw]
    else:
        im2_aug = im1.clone()
        im2_aug[..., cy:cy + ch, cx:cx + cw] = im2[..., cy:cy + ch, cx:cx + cw]
        im2 = im2_aug
    return im1, im2


def rgb(im1, im2, prob=1.0):
    if np.random.rand(1) >= prob:
        return im1, im2
    perm = np.random.permutation(3)
    return im1[:, :, perm, :, :], im2[:, :, perm, :, :]
