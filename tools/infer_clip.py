#!/usr/bin/env python
"""Clip inference with and without per-frame feature reuse (realvsr_amd.infer.SlidingWindowRunner).

Default = BASELINE config 5 geometry (EDVR nf128, 7 frames, 540x960 LR -> 2160x3840) on a T-frame clip; prints
ms per output frame for the reference's window-by-window loop and for the feature-reuse driver, checks that the
two outputs are bit-identical and converts one frame to BGR uint8 on the GPU."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--nf', type=int, default=128)
ap.add_argument('--nframes', type=int, default=7)
ap.add_argument('--height', type=int, default=540)
ap.add_argument('--width', type=int, default=960)
ap.add_argument('--T', type=int, default=10, help='clip length')
ap.add_argument('--padding', default='replicate')
a = ap.parse_args()

from realvsr_amd.archs.EDVR_arch import EDVR  # noqa: E402
from realvsr_amd.infer import SlidingWindowRunner, ycbcr_to_bgr_u8  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
net = EDVR(nf=a.nf, nc=3, nframes=a.nframes, groups=8, front_RBs=5, back_RBs=10, w_TSA=True)
gen = torch.Generator().manual_seed(99)
with torch.no_grad():
    for name, p in net.named_parameters():
        if 'conv_offset_mask.weight' in name:
            p.copy_(torch.randn(p.shape, generator=gen) * 0.01)
net = net.to(dev).eval()
clip = torch.rand(a.T, 3, a.height, a.width, generator=torch.Generator().manual_seed(1234)).to(dev)
run = SlidingWindowRunner(net, a.nframes, padding=a.padding, chunk=2)


def timed(fn):
    fn()  # warm-up (workspace, autotuned nothing; just first-touch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3 / a.T


ref, ms_ref = timed(lambda: run.reference_order(clip))
out, ms_reuse = timed(lambda: run(clip))
# fused-DCN forward launches of one more eager feature-reuse pass, HIP-event timed (same accounting as bench.py)
from bench import DcnTimer, HBM_PEAK_GBS  # noqa: E402
timer = DcnTimer()
timer.install()
run(clip)
torch.cuda.synchronize()
timer.uninstall()
nl, kms, kbytes = timer.result()
run_g = SlidingWindowRunner(net, a.nframes, padding=a.padding, chunk=2, use_graph=True)
out_g, ms_graph = timed(lambda: run_g(clip))
bgr = ycbcr_to_bgr_u8(out[0])
print(json.dumps({'metric': 'HR frames/sec (fwd only, sliding window) on %d-frame %dx%d LR windows' % (a.nframes, a.height, a.width),
                  'value': round(1e3 / min(ms_reuse, ms_graph), 3), 'unit': 'HR frames/s',
                  'roofline': {'kernel': 'dcn_fwd3_kernel (+ weight pre-pack), fused DCN forward', 'bound': 'hbm',
                               'achieved': round(kbytes / (kms * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': round(kbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'launches': nl,
                               'avg_launch_ms': round(kms / max(nl, 1), 4), 'algorithmic_bytes_per_launch': round(kbytes / max(nl, 1))},
                  'config': 'EDVR nf%d %df %dx%d x4, clip of %d frames, padding %s' % (a.nf, a.nframes, a.height, a.width, a.T, a.padding),
                  'ms_per_frame_window_by_window': round(ms_ref, 2), 'ms_per_frame_feature_reuse': round(ms_reuse, 2),
                  'ms_per_frame_feature_reuse_hipgraph': round(ms_graph, 2), 'speedup': round(ms_ref / ms_reuse, 3),
                  'bit_identical': bool(torch.equal(ref, out)), 'graph_bit_identical': bool(torch.equal(out_g, out)),
                  'bgr_u8_shape': list(bgr.shape), 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
