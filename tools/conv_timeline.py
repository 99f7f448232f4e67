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
buf = (ctypes.c_ulonglong * 512)()
print('rc', L.rvsr_debug_read(buf))
t = list(buf)
names = {0: 'tile start'}
V3 = True   # (round 3: conv_fwd5 is the only 3x3 stride-1 kernel)
V5 = True
if V3:
    # conv_fwd3_kernel: stamps of the last 8 slots of workgroup 77, wave 0 (group A: even slots MFMA, odd slots staging)
    names = {}
    for sl in range(8):
        kind = ('mfma + staging slices' if V5 else ('mfma(+epilogue)' if sl % 2 == 0 else 'commit+issue'))
        names[1 + 3 * sl] = 'slot%d start' % sl
        names[2 + 3 * sl] = 'slot%d %s done' % (sl, kind)
        names[3 + 3 * sl] = 'slot%d barrier passed' % sl
        if sl % 2 and not V5:
            names[40 + 2 * sl] = 'slot%d   loads landed' % sl
            names[41 + 2 * sl] = 'slot%d   committed to LDS' % sl
    print('workgroup 77: %d stages, kernel start -> end %d ticks (%.0f per stage)' % (t[62], t[61] - t[60], (t[61] - t[60]) / max(t[62], 1)))
    if V5:
        print('stage Q-3 per wave: loop time after the common barrier release:', [int(t[70 + w] - t[80 + w]) for w in range(8)])
        print('stage Q-4 loop-end skew per wave:', [int(t[90 + w] - min(t[90:98])) for w in range(8)])
        t0 = min(t[300 + w * 12 + 11] for w in range(8))
        print('stage Q-6, ticks since the first wave left the previous barrier; per wave: barrier passed | start of taps 0..8 | loop end | next barrier passed')
        for w in range(8):
            r = [int(t[300 + w * 12 + i] - t0) for i in (11, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10)]
            print('  wave %d:' % w, r, ' per tap:', [r[i + 1] - r[i] for i in range(1, 10)])
    order = sorted(names, key=lambda i: t[i])
    prev = t[order[0]]
    for i in order:
        print('%-32s +%7d' % (names[i], t[i] - prev))
        prev = t[i]
    sys.exit(0)
for c in range(4):
    names[1 + 5 * c] = 'chunk%d commit done' % c
    names[2 + 5 * c] = 'chunk%d barrier1 passed' % c
    names[3 + 5 * c] = 'chunk%d prefetch issued' % c
    names[4 + 5 * c] = 'chunk%d mfma done' % c
    names[5 + 5 * c] = 'chunk%d barrier2 passed' % c
names[30] = 'epilogue issued'
names[31] = 'next tile prefetch issued'
prev = t[0]
for i in sorted(names):
    print('%-28s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
