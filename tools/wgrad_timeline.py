"""s_memtime timeline of conv_wgrad2_kernel (workgroup 77, wave 0, first six tiles): build with tools/build_variant_conv.sh tl -DRVSR_TIMELINE,
run with RVSR_SO=<that library> on the GPU box."""
import ctypes, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
conv = nn.Conv2d(64, 64, 3, 1, 1).to(dev)
x = torch.randn(40, 64, 180, 320, device=dev)
for _ in range(3):
    conv.weight.grad = None
    y = RF.conv2d(x, conv, RF.ACT_LRELU)
    y.backward(torch.randn_like(y))
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 512)()
print('rc', L.rvsr_debug_read(buf))
t = list(buf)
for ti in range(6):
    s = t[200 + ti * 5:205 + ti * 5]
    nxt = t[200 + (ti + 1) * 5] if ti < 5 else None
    print('tile %d: commit %6d | barrier %6d | MFMA phase (+ next tile\'s loads) %6d | closing barrier %s' %
          (ti, s[1] - s[0], s[2] - s[1], s[4] - s[3], (nxt - s[4]) if nxt else '-'))
t0 = min(t[120 + w * 10] for w in range(8))
print('tile 3, per wave: start of k-steps 0..7 | loop end (ticks since the first wave entered the loop); per k-step')
for w in range(8):
    r = [int(t[120 + w * 10 + i] - t0) for i in range(9)]
    print('  wave %d:' % w, r, [r[i + 1] - r[i] for i in range(8)])
