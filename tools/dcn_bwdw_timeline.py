"""s_memtime timeline of one dcn_bwdw3_kernel tile (build: tools/build_timeline.sh; RVSR_SO=...librealvsr_tl.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(40, 64, 180, 320, generator=g).to(dev).requires_grad_(True)
om = torch.randn(40, 216, 180, 320, generator=g); om[:, :144] *= 0.1; om = om.to(dev).requires_grad_(True)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev).requires_grad_(True); b = torch.zeros(64, device=dev, requires_grad=True)
gout = torch.randn(40, 64, 180, 320, generator=g).to(dev)
for _ in range(2):
    out = RF.dcn_pack(x, om, w, b, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
    out.backward(gout)
torch.cuda.synchronize()
L = ctypes.CDLL(os.environ['RVSR_SO'])
buf = (ctypes.c_ulonglong * 256)()
print('rc', L.rvsr_debug_read_dcn(buf))
t = list(buf)
nm = {180: 'tile start', 181: 'gOut tile staged (loads + convert + LDS)', 182: 'x tile staged', 183: 'barrier',
      184: 'offsets/masks of 3 items loaded (issue)', 185: 'column tile built', 186: 'barrier', 187: '18 bf16 MFMAs', 188: 'barrier'}
prev = t[180]
for i in sorted(nm):
    print('%-44s +%7d' % (nm[i], t[i] - prev))
    prev = t[i]
