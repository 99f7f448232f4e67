"""Host time needed to ENQUEUE one optimizer step (no synchronisation inside the measured call) next to the GPU time of the step:
how far the step is from being launch-bound."""
import sys, os, time, argparse, torch
sys.path.insert(0, os.getcwd())
import bench
from realvsr_amd import VideoSR_model
args = argparse.Namespace(nf=64, nframes=5, back_rbs=10)
dev = torch.device('cuda:0')
model = VideoSR_model.create_model(bench.model_opt(args, 1))
bench.init_weights(model.netG)
x, gt = bench.make_batch(8, 5, 180, 320, dev)
model.feed_data({'LQs': x, 'GT': gt})
for i in range(3):
    model.optimize_parameters(i + 1, log=False)
torch.cuda.synchronize()
import gc
if os.environ.get('NOGC'): gc.disable()
cpu, tot = [], []
for i in range(24):
    na = torch.cuda.memory_stats().get('num_device_alloc', 0); g2 = gc.get_stats()[2]['collections']
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.optimize_parameters(10 + i, log=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    cpu.append(1e3 * (t1 - t0)); tot.append(1e3 * (t2 - t0))
    if cpu[-1] > 20: print('slow step', i, 'device allocs during it:', torch.cuda.memory_stats().get('num_device_alloc', 0) - na, 'gen2 collections:', gc.get_stats()[2]['collections'] - g2)
print('host enqueue ms per step:', ['%.1f' % v for v in cpu], ' step incl. GPU:', ['%.1f' % v for v in tot])
