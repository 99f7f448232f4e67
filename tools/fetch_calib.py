#!/usr/bin/env python
"""Calibration of rocprofv3 FETCH_SIZE for the access pattern our kernels use (4 B per lane,
lanes along W, fully coalesced): charb_fwd_kernel reads exactly 2 * n * 4 bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF  # noqa: E402

n = 96 * 1024 * 1024  # 2 x 384 MiB: larger than the 256 MiB Infinity Cache
x = torch.rand(n, device='cuda')
y = torch.rand(n, device='cuda')
for _ in range(3):
    v = RF.charbonnier(x, y)
torch.cuda.synchronize()
print('charbonnier over 2 x %d floats = %d bytes read' % (n, 2 * n * 4), float(v))
