import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 64, 180, 320, generator=g).to(dev)
om = torch.randn(8, 216, 180, 320, generator=g); om[:, :144] *= 0.1; om = om.to(dev)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev); b = torch.zeros(64, device=dev)
for _ in range(3):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read_dcn(buf))
t = list(buf)
names = {0: 'start'}
for c in range(4):
    names[1 + 5 * c] = 'chunk%d weights copied' % c
    names[2 + 5 * c] = 'chunk%d x tile staged' % c
    names[3 + 5 * c] = 'chunk%d barrier1' % c
    names[4 + 5 * c] = 'chunk%d 9 taps (build+mfma)' % c
    names[5 + 5 * c] = 'chunk%d barrier2' % c
names[30] = 'epilogue'
prev = t[0]
for i in sorted(names):
    print('%-30s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
