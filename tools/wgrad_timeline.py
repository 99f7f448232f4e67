import ctypes, os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
conv = nn.Conv2d(64, 64, 3, 1, 1).to(dev)
x = torch.randn(40, 64, 180, 320, device=dev)
gout = torch.randn(40, 64, 180, 320, device=dev)
for _ in range(2):
    conv.zero_grad()
    y = RF.conv2d(x, conv, RF.ACT_LRELU)
    y.backward(gout)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read(buf))
t = list(buf)
lab = ['loop top', 'commit done', 'barrier1', 'next loads issued', 'mfma done']
prev = t[100]
for ti in range(6):
    for k in range(5):
        i = 100 + ti * 5 + k
        print('tile%d %-18s +%7d (t=%d)' % (ti, lab[k], t[i] - prev, t[i] - t[100]))
        prev = t[i]
