"""Which torch ops (not our kernels) still run in one training step, with shapes and autograd scope."""
import os, sys, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench
ap = argparse.Namespace(nf=64, nframes=5, back_rbs=10, batch=8, height=180, width=320)
from realvsr_amd.VideoSR_model import create_model
torch.cuda.set_device(0)
torch.manual_seed(0)
model = create_model(bench.model_opt(ap, 1))
bench.init_weights(model.netG)
x, gt = bench.make_batch(8, 5, 180, 320, torch.device('cuda:0'))
model.feed_data({'LQs': x, 'GT': gt})
for i in range(2):
    model.optimize_parameters(i + 1, log=False)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    model.optimize_parameters(3, log=False)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_time_total > 100 and e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::empty_like', 'aten::zeros_like', 'aten::view'):
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:40]:
    print('%8.0f us  %-28s %s' % r)
