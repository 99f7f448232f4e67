"""s_memtime timeline of one dcn_fwd4_kernel workgroup, per wave (build: TLFLAGS=-DRVSR_TIMELINE_DCN4 tools/build_timeline.sh;
run with RVSR_DCN_FWD=4 RVSR_SO=$PWD/realvsr_amd/csrc/librealvsr_tl.so python tools/dcn4_timeline.py [offset std])."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
ostd = float(sys.argv[1]) if len(sys.argv) > 1 else 1.25
x = torch.randn(40, 64, 180, 320, generator=g).to(dev).requires_grad_(True)
om = torch.randn(40, 216, 180, 320, generator=g); om[:, :144] *= ostd; om = om.to(dev)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev); b = torch.zeros(64, device=dev)
for _ in range(3):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 1024)()
print('rc', L.rvsr_debug_read_dcn4(buf), 'offset std', ostd)
t = list(buf)
nw = 8   # waves per workgroup (Fwd4<8, 5, 7, MT>)
t0 = min(t[w * 64] for w in range(nw) if t[w * 64])
for w in range(nw):
    s = t[w * 64:(w + 1) * 64]
    print('w%-2d start %6d prologue %6d | tile 2 begins %8d ... epilogue done %8d (tile: %6d cycles) | kernel end %9d' % (w, s[0] - t0, s[1] - t0, s[8] - t0, s[57] - t0, s[57] - s[8], s[58] - t0))
for w in (0, nw - 1):
    s = t[w * 64:(w + 1) * 64]
    for p in range(4):
        b = 8 + 12 * p
        print('w%-2d period %d: iterations' % (w, p), [s[b + k] - s[b + k - 1] for k in range(1, 10)], 'fix-up check %5d' % (s[b + 10] - s[b + 9]),
              'to next body %6d' % ((s[b + 12] if p < 3 else s[57]) - s[b + 10]))
