"""s_memtime timeline of one wave (wave 3 of workgroup 77) of dcn_bwdin6_kernel (batch element 1) and of dcn_bwdw6_kernel (third tile of its stream) at the L1 shape.
   tools/build_variant.sh tl6 dcn6_kernels.hip -DRVSR_TIMELINE_DCN6 ; RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl6.so python tools/dcn6_timeline.py [ostd]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
ostd = float(sys.argv[1]) if len(sys.argv) > 1 else 1.25
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
print('rc', L.rvsr_debug_read_dcn6(buf))
t = list(buf)
names = {0: 'prologue done (gOut fragments, gOut^T emitted, window zeroed, chunk 0 requested)', 9: 'kernel end'}
for c in range(4):
    names[10 + 8 * c] = 'chunk%d top' % c
    names[11 + 8 * c] = 'chunk%d x tile committed' % c
    names[12 + 8 * c] = 'chunk%d vmcnt(0)' % c
    names[13 + 8 * c] = 'chunk%d barrier' % c
    names[14 + 8 * c] = 'chunk%d lane iterations done' % c
    names[15 + 8 * c] = 'chunk%d barrier' % c
    names[16 + 8 * c] = 'chunk%d next requests issued' % c
    names[17 + 8 * c] = 'chunk%d flush done' % c
for it in range(5):
    names[50 + 4 * it] = '  chunk1 iteration %d start' % it
    names[51 + 4 * it] = '  chunk1 iteration %d reads + math + atomics issued' % it
for mt in range(3):
    names[90 + mt] = '  chunk1 M tile %d MFMAs issued' % mt
print('--- dcn_bwdin6')
prev = t[0]
for i in sorted(names, key=lambda i: t[i]):
    if t[i] == 0:
        continue
    print('%-84s +%7d  (t=%d)' % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
print('--- dcn_bwdw6 (third tile of one stream)')
nm = {0: 'tile top', 1: 'vmcnt(0): operands + x landed', 2: 'x committed + barrier', 3: 'next requests issued', 4: 'iterations done', 5: 'end barrier'}
for it in range(5):
    nm[10 + 4 * it] = '  iteration %d start' % it
    nm[11 + 4 * it] = '  iteration %d column values blended' % it
    nm[12 + 4 * it] = '  iteration %d split + transposer issued' % it
u = t[128:]
prev = u[0]
for i in sorted(nm, key=lambda i: u[i]):
    if u[i] == 0:
        continue
    print('%-84s +%7d  (t=%d)' % (nm[i], u[i] - prev, u[i] - u[0]))
    prev = u[i]
