"""Where do conv_fwd7 and conv_fwd5 differ?  (developer diagnostic)"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, C, Co, H, W) in [(1, 64, 64, 8, 64), (1, 64, 64, 16, 128), (3, 64, 64, 180, 320), (1, 32, 64, 8, 64)]:
    conv = nn.Conv2d(C, Co, 3, 1, 1).to(dev)
    x = torch.randn(B, C, H, W, device=dev)
    outs = []
    for sw in ('0', '1'):
        os.environ['RVSR_CONV_FWD7'] = sw
        with torch.no_grad():
            o = torch.full((B, Co, H, W), 7777.0, device=dev)
            y = RF.conv2d(x, conv, RF.ACT_NONE)
            outs.append(y.clone())
    torch.cuda.synchronize()
    a, b = outs
    bad = (a != b)
    print((B, C, Co, H, W), 'differing elements', int(bad.sum()), 'of', bad.numel())
    if bad.any():
        idx = bad.nonzero()
        print('  channels:', sorted(set(idx[:, 1].tolist()))[:70])
        print('  rows % 8:', sorted(set((idx[:, 2] % 8).tolist())), ' rows:', sorted(set(idx[:, 2].tolist()))[:40])
        print('  cols % 64:', sorted(set((idx[:, 3] % 64).tolist()))[:70])
        print('  batch:', sorted(set(idx[:, 0].tolist())))
        i0 = idx[0].tolist()
        print('  first', i0, 'fwd5', a[tuple(i0)].item(), 'fwd7', b[tuple(i0)].item())
        # is the fwd7 value some OTHER element of fwd5's output? (a permutation bug)
        v = b[tuple(i0)]
        where = (a == v).nonzero()
        print('  fwd7 value found in fwd5 output at', where[:4].tolist())
