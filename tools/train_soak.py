#!/usr/bin/env python
"""Soak: N optimizer steps of the bench's training step (config 2 geometry by default) on ONE fixed synthetic batch, in a GEMM mode; prints the
loss every `--every` steps.  The curves of the speed modes against the default mode, and the absence of non-finite values over a few hundred
steps, are the evidence (profiles/r04_notes.md)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--every', type=int, default=25)
ap.add_argument('--mode', default='bf16x3')
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--lr', type=float, default=2e-4)
ap.add_argument('--seed', type=int, default=0)
a = ap.parse_args()
from realvsr_amd import _lib as rlib  # noqa: E402
from realvsr_amd.VideoSR_model import create_model  # noqa: E402
rlib.set_gemm_mode(a.mode)
cfg = bench._Cfg(nf=64, nframes=5, back_rbs=10, height=180, width=320, batch=a.batch, lf_mode='ssim', offset_px=None)
torch.manual_seed(a.seed)
opt = bench.model_opt(cfg, 1)
opt['train']['lr_G'] = a.lr
model = create_model(opt)
bench.init_weights(model.netG)
x, gt = bench.make_batch(a.batch, 5, 180, 320, torch.device('cuda:0'))
model.feed_data({'LQs': x, 'GT': gt})
curve = []
for step in range(1, a.steps + 1):
    log = step % a.every == 0 or step == 1
    model.optimize_parameters(step, log=log)
    if log:
        l = model.get_current_log()
        curve.append((step, round(l['l_pix'], 6)))
flat = model.optimizer_G.buffers.param
print(json.dumps({'mode': a.mode, 'seed': a.seed, 'steps': a.steps, 'lr': a.lr, 'batch': a.batch, 'loss': curve,
                  'params_finite': bool(torch.isfinite(flat).all().item())}))
