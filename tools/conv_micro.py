#!/usr/bin/env python
"""Micro-benchmark of the fused 3x3 conv block (64->64, 40 frames of 180x320 = the EDVR-M feature
extraction shape of BASELINE config 2); for rocprofv3 kernel-trace / PMC passes."""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--B', type=int, default=40)
ap.add_argument('--C', type=int, default=64)
ap.add_argument('--Co', type=int, default=64)
ap.add_argument('--H', type=int, default=180)
ap.add_argument('--W', type=int, default=320)
ap.add_argument('--bwd', action='store_true')
ap.add_argument('--no-act', action='store_true', help='no activation: the weight / data gradient kernels without the act mask')
a = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
conv = nn.Conv2d(a.C, a.Co, 3, 1, 1).to(dev)
x = torch.randn(a.B, a.C, a.H, a.W, device=dev, requires_grad=a.bwd)
gout = torch.randn(a.B, a.Co, a.H, a.W, device=dev)
for it in range(a.iters + 1):
    if it == 1:
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    y = RF.conv2d(x, conv, RF.ACT_NONE if a.no_act else RF.ACT_LRELU)
    if a.bwd:
        y.backward(gout)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / a.iters
flop = 2.0 * a.C * 9 * a.Co * a.B * a.H * a.W * (3 if a.bwd else 1)
if os.environ.get('RVSR_MICRO_CHECK'):
    import torch.nn.functional as F
    with torch.no_grad():
        ref = F.conv2d(x[:2].double(), conv.weight.double(), conv.bias.double(), padding=1)
        if not a.no_act:
            ref = F.leaky_relu(ref, 0.1)
        err = (y[:2].double() - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()
    print('  l2 error of the output against an f64 conv: %.2e' % err.item())
if os.environ.get('RVSR_MICRO_HASH'):
    print('  output hash %d' % int(y.detach().view(torch.int32).to(torch.int64).sum().item()))
print('conv %s: %.3f ms/iter, %.1f TFLOP/s (f32-equivalent), %.3f ns/px' % ('fwd+bwd' if a.bwd else 'fwd', ms, flop / ms / 1e9,
                                                                              ms * 1e6 / (a.B * a.H * a.W)))
