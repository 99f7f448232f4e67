"""Per-tap s_memtime timeline of conv_fwd7_kernel (build: tools/build_variant.sh conv7_kernels tl7 -DRVSR_TIMELINE7; run with
RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl7.so).  Third tile of workgroup 77, all eight waves, 40 x 64 x 180 x 320."""
import ctypes, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
conv = nn.Conv2d(64, 64, 3, 1, 1).to(dev)
x = torch.randn(40, 64, 180, 320, device=dev)
for _ in range(3):
    y = RF.conv2d(x, conv, RF.ACT_LRELU)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * (8 * 96))()
print('rc', L.rvsr_debug_read7(buf))
t = [list(buf[w * 96:(w + 1) * 96]) for w in range(8)]
# stamps per stage: FIRST: start, 9 taps (pass A), pass-B start, 9 taps, end, after barrier = 22; MID: start, 9 taps, end, after barrier = 12;
# LAST: start, 9 taps, pass-A end, parked, 9 taps, end, after barrier = 23
layout = [('FIRST', ['start'] + ['A%d' % i for i in range(9)] + ['A end'] + ['B%d' % i for i in range(9)] + ['end', 'barrier']),
          ('MID1', ['start'] + ['t%d' % i for i in range(9)] + ['end', 'barrier']),
          ('MID2', ['start'] + ['t%d' % i for i in range(9)] + ['end', 'barrier']),
          ('LAST', ['start'] + ['A%d' % i for i in range(9)] + ['A end', 'parked'] + ['B%d' % i for i in range(9)] + ['end', 'barrier'])]
t0 = min(t[w][0] for w in range(8))
i = 0
for name, labels in layout:
    print('%s (ticks since the tile started; per wave: duration of each segment)' % name)
    for w in range(8):
        seg = [int(t[w][i + j + 1] - t[w][i + j]) for j in range(len(labels) - 1)]
        print('  wave %d @%6d:' % (w, t[w][i] - t0), ' '.join('%s %d' % (labels[j], seg[j]) for j in range(len(seg))))
    i += len(labels)
print('tile total per wave:', [int(t[w][i - 1] - t[w][0]) for w in range(8)])
