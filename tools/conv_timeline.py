import ctypes, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF, _lib
dev = torch.device('cuda:0')
conv = nn.Conv2d(64, 64, 3, 1, 1).to(dev)
x = torch.randn(40, 64, 180, 320, device=dev)
for _ in range(3):
    y = RF.conv2d(x, conv, RF.ACT_LRELU)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read(buf))
t = list(buf)
names = {0: 'tile start'}
V3 = os.environ.get('RVSR_CONV_FWD', '3') != '2'
for c in range(4):
    if V3:
        names[1 + 5 * c] = 'chunk%d next-stage loads issued' % c
        names[2 + 5 * c] = 'chunk%d mfma done' % c
        names[3 + 5 * c] = 'chunk%d next stage committed' % c
        names[4 + 5 * c] = 'chunk%d barrier passed' % c
    else:
        names[1 + 5 * c] = 'chunk%d commit done' % c
        names[2 + 5 * c] = 'chunk%d barrier1 passed' % c
        names[3 + 5 * c] = 'chunk%d prefetch issued' % c
        names[4 + 5 * c] = 'chunk%d mfma done' % c
        names[5 + 5 * c] = 'chunk%d barrier2 passed' % c
names[30] = 'epilogue issued'
if not V3:
    names[31] = 'next tile prefetch issued'
prev = t[0]
for i in sorted(names):
    print('%-28s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
