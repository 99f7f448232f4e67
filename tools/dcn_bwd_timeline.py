import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 64, 180, 320, generator=g).to(dev).requires_grad_(True)
om = torch.randn(8, 216, 180, 320, generator=g); om[:, :144] *= 0.1; om = om.to(dev).requires_grad_(True)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
gout = torch.randn(8, 64, 180, 320, generator=g).to(dev)
for _ in range(2):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
    out.backward(gout)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read_dcn(buf))
t = list(buf)
names = {100: 'start', 101: 'gOut fragments loaded', 102: 'private windows zeroed'}
for c in range(3):
    names[103 + 5 * c] = 'chunk%d offsets+weights issued/copied' % c
    names[104 + 5 * c] = 'chunk%d x tile staged' % c
    names[105 + 5 * c] = 'chunk%d barrier' % c
    names[106 + 5 * c] = 'chunk%d 3 M tiles (mfma+consume+scatter)' % c
    names[107 + 5 * c] = 'chunk%d barrier + merge' % c
names[130] = 'end (8 chunks)'
for tp in range(4):
    names[140 + 4 * tp] = '  chunk1 tap%d start' % tp
    names[141 + 4 * tp] = '  chunk1 tap%d sampled + consumer math done' % tp
    names[142 + 4 * tp] = '  chunk1 tap%d claim/scatter done' % tp
    names[143 + 4 * tp] = '  chunk1 tap%d far-offset path passed' % tp
prev = t[100]
for i in sorted(names, key=lambda i: t[i]):
    print('%-44s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[100]))
    prev = t[i]
print('--- dcn_bwdw2_kernel, third tile of one workgroup (8 waves, 128 px, one 8-channel group)')
nm = {160: 'tile start', 161: 'gOut tile staged (4 batches of 4 loads)', 162: 'x tile staged', 163: 'barrier',
      164: 'offsets/masks of 3 items loaded (issue)', 165: 'column tile built', 166: 'barrier', 167: '48 f32 MFMAs', 168: 'barrier'}
prev = t[160]
for i in sorted(nm):
    print('%-44s +%7d' % (nm[i], t[i] - prev))
    prev = t[i]
