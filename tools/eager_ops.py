#!/usr/bin/env python
"""Which eager-torch kernels still run inside a training step (config 2), by operator and input shape: torch.profiler over 3 steps of bench.py's model."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from realvsr_amd.VideoSR_model import create_model  # noqa: E402
a = bench._Cfg(nf=64, nframes=5, back_rbs=10, batch=8, height=180, width=320, lf_mode='ssim', force_allreduce=False)
torch.manual_seed(0)
model = create_model(bench.model_opt(a, 1))
bench.init_weights(model.netG)
dev = torch.device('cuda:0')
x, gt = bench.make_batch(8, 5, 180, 320, dev)
bench.offset_stats(model.netG, x, 1.0)
model.feed_data({'LQs': x, 'GT': gt})
for i in range(3):
    model.optimize_parameters(i + 1, log=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(3):
        model.optimize_parameters(i + 4, log=False)
    torch.cuda.synchronize()
import time
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
busy = sum(e.time_range.end - e.time_range.start for e in evs) / 3.0
span = (max(e.time_range.end for e in evs) - min(e.time_range.start for e in evs)) / 3.0
print('device kernels: %d per step, busy %.2f ms/step of a %.2f ms/step span (idle %.1f %%)' % (len(evs) / 3, busy / 1e3, span / 1e3, 100 * (1 - busy / span)))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
tot = 0.0
for e in rows[:40]:
    tot += e.device_time_total
    print('%-28s calls/step %5.1f  us/step %8.1f  shapes %s' % (e.key, e.count / 3.0, e.device_time_total / 3.0, str(e.input_shapes)[:150]))
print('aten total us/step: %.1f' % (sum(e.device_time_total for e in rows) / 3.0))
