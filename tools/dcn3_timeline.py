"""s_memtime timeline of one dcn_fwd3_kernel workgroup (build: tools/build_variant.sh tl3 dcn3_kernels.hip -DRVSR_TIMELINE_DCN; run with RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl3.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
ostd = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
x = torch.randn(40, 64, 180, 320, generator=g).to(dev)
om = torch.randn(40, 216, 180, 320, generator=g); om[:, :144] *= ostd; om = om.to(dev)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev); b = torch.zeros(64, device=dev)
for _ in range(3):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read_dcn3(buf), 'offset std', ostd)
t = list(buf)
names = {0: 'start'}
for c in range(4):
    names[1 + 6 * c] = 'chunk%d loads issued' % c
    names[2 + 6 * c] = 'chunk%d weights in LDS' % c
    names[3 + 6 * c] = 'chunk%d x tile in LDS' % c
    names[4 + 6 * c] = 'chunk%d barrier1' % c
    names[5 + 6 * c] = 'chunk%d 9 taps' % c
    names[6 + 6 * c] = 'chunk%d barrier2' % c
names[30] = 'epilogue'
prev = t[0]
for i in sorted(names):
    print('%-30s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
print('chunk1 taps:', [t[40 + k] - (t[4 + 6] if k == 0 else t[39 + k]) for k in range(9)])
print('chunk1, inside each tap (cycles): offsets consumed + geometry + corner reads issued | corner reads back | blend (+ far path) | split | fragment reads + MFMAs issued')
for k in range(9):
    a = [t[60 + 5 * k + j] for j in range(4)] + [t[40 + k]]
    print('  tap %d: %s' % (k, ' | '.join('%5d' % (a[j + 1] - a[j]) for j in range(4))), ' (tap start at t=%d)' % (a[0] - t[0]))
