"""Training-time augmentation of an LQ/GT clip pair on the device.

Replaces codes/data/augments_video_allpair.py (apply_augment :6-35, blend :38-51, cutblur :54-77, rgb :80-88), which
the reference calls on GPU tensors right before netG (VideoSR_AllPair_model_YCbCr_Split.py:169-173).  There every
augmentation is a chain of clone / slice-assign / index / repeat ops (2-4 passes over both clips, more for blend);
here the step is split into

  * ``draw_plan`` -- the random DECISIONS (which augmentation, whether it fires, box, permutation, blend weight),
    drawn on the host from numpy's global RNG in exactly the reference's call order, so a seeded run makes the same
    choices; a plan is a small record, no tensor work;
  * ``apply_plan`` -- ONE HIP kernel (rvsr_augment_clips, csrc/train_kernels.hip) that reads both clips once and
    writes both augmented clips once, whatever the plan.

``apply_augment / blend / cutblur / rgb`` keep the reference's names and signatures on top of those two.
Quirk kept on purpose: cutblur sizes its box from ``im2.size(2), im2.size(3)`` (:64) -- for a 5-D clip
[B, N, 3, H, W] that is (3, H), not (H, W) -- and pastes it into the last two dimensions.
"""
import numpy as np
import torch

from . import functional as RF


class AugPlan:
    """What to do to the pair: channel permutation, box paste, blend.  ``fired`` is False when the draw decided to
    leave the clips alone (the reference then returns its inputs)."""
    __slots__ = ('fired', 'perm', 'box_mode', 'box', 'v', 'blend')

    def __init__(self):
        self.fired, self.perm, self.box_mode, self.box, self.v, self.blend = False, (0, 1, 2), 0, (0, 0, 0, 0), 1.0, False


def _draw_blend(plan, prob, alpha):
    if alpha <= 0 or np.random.rand(1) >= prob:
        return
    plan.fired, plan.blend = True, True      # the colour itself is drawn in apply_plan (torch RNG, as the reference)
    plan.v = float(np.random.uniform(alpha, 1))


def _draw_cutblur(plan, size, prob, alpha, size1=None):
    if size1 is not None and tuple(size1) != tuple(size):   # checked BEFORE any draw, as the reference does (:55-56)
        raise ValueError('im1 and im2 have to be the same resolution.')
    if alpha <= 0 or np.random.rand(1) >= prob:
        return
    cut_ratio = np.random.randn() * 0.01 + alpha
    h, w = size[2], size[3]
    ch, cw = int(h * cut_ratio), int(w * cut_ratio)
    cy = np.random.randint(0, h - ch + 1)
    cx = np.random.randint(0, w - cw + 1)
    inside = np.random.random() > 0.5
    plan.fired = True
    plan.box_mode = 1 if inside else 2
    # python slice semantics of [cy:cy+ch, cx:cx+cw] on the last two dimensions: clamped, and a NEGATIVE stop (alpha near 0
    # makes ch / cw negative) counts from the end, exactly as the reference's slice assignment does
    H, W = size[-2], size[-1]
    ys, ye, _ = slice(cy, cy + ch).indices(H)
    xs, xe, _ = slice(cx, cx + cw).indices(W)
    plan.box = (ys, max(ye, ys), xs, max(xe, xs))


def _draw_rgb(plan, prob):
    if np.random.rand(1) >= prob:
        return
    plan.fired = True
    plan.perm = tuple(int(i) for i in np.random.permutation(3))


def draw_plan(size, augs, probs, alphas, mix_p=None, size1=None):
    """Host-side draw for clips of shape ``size`` (im2; ``size1`` = im1's shape when known); numpy RNG call order =
    apply_augment + the chosen augmentation."""
    idx = np.random.choice(len(augs), p=mix_p)
    aug, prob, alpha = augs[idx], float(probs[idx]), float(alphas[idx])
    plan = AugPlan()
    if aug == 'none':
        pass
    elif aug == 'blend':
        _draw_blend(plan, prob, alpha)
    elif aug == 'cutblur':
        _draw_cutblur(plan, size, prob, alpha, size1)
    elif aug == 'rgb':
        _draw_rgb(plan, prob)
    else:
        raise ValueError('{} is not invalid.'.format(aug))
    return plan


def apply_plan(im1, im2, plan):
    """Both augmented clips from one pass over both inputs.  Always returns new tensors (the reference's callers get
    clones / fresh tensors from apply_augment as well)."""
    same = im1.shape == im2.shape
    if plan.box_mode and not same:
        raise ValueError('im1 and im2 have to be the same resolution.')
    colour = None
    if plan.blend:
        colour = torch.empty((im2.size(0), im2.size(1), 3, 1, 1), device=im2.device).uniform_(0, 1)
    if same:
        return RF.augment_clips(im1, im2, plan.perm, plan.box_mode, plan.box, plan.v, colour)
    # x4 models: LQ and GT clips differ in size; only the size-agnostic augmentations reach here
    out1, _ = RF.augment_clips(im1, im1, plan.perm, 0, (0, 0, 0, 0), plan.v, colour)
    out2, _ = RF.augment_clips(im2, im2, plan.perm, 0, (0, 0, 0, 0), plan.v, colour)
    return out1, out2


def apply_augment(im1, im2, augs, probs, alphas, mix_p=None):
    return apply_plan(im1, im2, draw_plan(tuple(im2.shape), augs, probs, alphas, mix_p, size1=tuple(im1.shape)))


def blend(im1, im2, prob=1.0, alpha=0.6):
    plan = AugPlan()
    _draw_blend(plan, prob, alpha)
    return apply_plan(im1, im2, plan) if plan.fired else (im1, im2)


def cutblur(im1, im2, prob=1.0, alpha=1.0):
    if im1.size() != im2.size():
        raise ValueError('im1 and im2 have to be the same resolution.')
    plan = AugPlan()
    _draw_cutblur(plan, tuple(im2.shape), prob, alpha)
    return apply_plan(im1, im2, plan) if plan.fired else (im1, im2)


def rgb(im1, im2, prob=1.0):
    plan = AugPlan()
    _draw_rgb(plan, prob)
    return apply_plan(im1, im2, plan) if plan.fired else (im1, im2)
