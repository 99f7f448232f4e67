"""s_memtime timeline of one wave (wave 3 of workgroup 77, batch element 1) of dcn_bwdin5_kernel at the L1 shape.
   tools/build_variant.sh tl5 dcn5_kernels.hip -DRVSR_TIMELINE_DCN5 ; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl5.so python tools/dcn5_timeline.py [ostd]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
ostd = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 64, 180, 320, generator=g).to(dev).requires_grad_(True)
om = torch.randn(8, 216, 180, 320, generator=g); om[:, :144] *= ostd; om = om.to(dev).requires_grad_(True)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
gout = torch.randn(8, 64, 180, 320, generator=g).to(dev)
for _ in range(2):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
    out.backward(gout)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read_dcn5(buf))
t = list(buf)
names = {0: 'prologue done (gOut fragments, norms, window zeroed)', 1: 'chunk 0 requests issued'}
for c in range(4):
    names[10 + 8 * c] = 'chunk%d top' % c
    names[11 + 8 * c] = 'chunk%d x tile committed' % c
    names[12 + 8 * c] = 'chunk%d vmcnt(0)' % c
    names[13 + 8 * c] = 'chunk%d barrier' % c
    names[14 + 8 * c] = 'chunk%d taps done' % c
    names[15 + 8 * c] = 'chunk%d barrier' % c
    names[16 + 8 * c] = 'chunk%d next requests issued' % c
    names[17 + 8 * c] = 'chunk%d flush done' % c
for tp in range(9):
    names[50 + 4 * tp] = '  chunk1 tap%d start' % tp
    names[51 + 4 * tp] = '  chunk1 tap%d corner reads + math done' % tp
    names[52 + 4 * tp] = '  chunk1 tap%d atomics issued' % tp
for mt in range(3):
    names[90 + mt] = '  chunk1 M tile %d MFMAs issued' % mt
prev = t[0]
for i in sorted(names, key=lambda i: t[i]):
    if t[i] == 0:
        continue
    print('%-52s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
