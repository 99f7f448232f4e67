#!/usr/bin/env python
"""Micro-benchmark of the fused DCN pack (forward + backward) at the EDVR L1 shape of BASELINE
config 2 (B=8, C=Co=64, dg=8, 180x320); used for rocprofv3 kernel-trace / PMC passes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--B', type=int, default=8)
ap.add_argument('--C', type=int, default=64)
ap.add_argument('--H', type=int, default=180)
ap.add_argument('--W', type=int, default=320)
ap.add_argument('--ostd', type=float, default=1.0, help='std of the offsets in pixels')
ap.add_argument('--fwd-only', action='store_true')
ap.add_argument('--smooth', type=int, default=0, help='offset field = noise on a grid S x coarser, bilinearly upsampled and rescaled to --ostd '
                '(0: i.i.d. per pixel and tap, the harshest case for the LDS gathers)')
ap.add_argument('--coherent', type=float, default=None, help='all offsets equal to this value (no sub-pixel sign jitter between neighbouring pixels)')
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(a.B, a.C, a.H, a.W, generator=g).to(dev).requires_grad_(True)
om = torch.randn(a.B, 216, a.H, a.W, generator=g)
if a.smooth > 1:
    import torch.nn.functional as F
    coarse = torch.randn(a.B, 144, (a.H + a.smooth - 1) // a.smooth + 1, (a.W + a.smooth - 1) // a.smooth + 1, generator=g)
    fine = F.interpolate(coarse, size=(a.H, a.W), mode='bilinear', align_corners=True)
    om[:, :144] = fine / fine.std()
om[:, :144] *= a.ostd
if a.coherent is not None:
    om[:, :144] = a.coherent
om = om.to(dev).requires_grad_(True)
w = (torch.randn(a.C, a.C, 3, 3, generator=g) / 24).to(dev).requires_grad_(True)
b = torch.zeros(a.C, device=dev, requires_grad=True)
gout = torch.randn(a.B, a.C, a.H, a.W, generator=g).to(dev)
for it in range(a.iters + 1):
    if it == 1:
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
    if not a.fwd_only:
        out.backward(gout)
e.record()
torch.cuda.synchronize()
px = a.B * a.H * a.W
print('dcn pack %s: %.3f ms/iter, %.3f ns/px' % ('fwd' if a.fwd_only else 'fwd+bwd', s.elapsed_time(e) / a.iters,
                                                s.elapsed_time(e) / a.iters * 1e6 / px))
