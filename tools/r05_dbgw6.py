#!/usr/bin/env python
"""round-5 debug aid (needs the -DRVSR_ABLW6=64 build): bisect a run-to-run difference of the DCN weight gradient."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF, _lib
L = _lib.lib()
g = torch.Generator().manual_seed(0)
dev = 'cuda:0'
B, C, Co, dg, H, W, ostd = 1, 64, 64, 8, 180, 320, 0.3
x = torch.randn(B, C, H, W, generator=g).to(dev)
om = torch.randn(B, dg * 27, H, W, generator=g)
om[:, :dg * 18] *= ostd
om = om.to(dev)
w = (torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev).requires_grad_(True)
b = torch.randn(Co, generator=g).to(dev)
gout = torch.randn(B, Co, H, W, generator=g).to(dev)
runs = []
for i in range(12):
    ls = [x.clone().requires_grad_(True), om.clone().requires_grad_(True), w, b.clone().requires_grad_(True)]
    w.grad = None
    RF.dcn_pack(*ls, 1, 1, 1, dg, 0, 0.1).backward(gout)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 256 * 8, dtype=np.float32)
    rc = L.rvsr_debug_read_w6(buf.ctypes.data_as(ctypes.c_void_p))
    runs.append((w.grad.clone().cpu().numpy(), buf.reshape(512, 256, 8).copy()))
ref = runs[0]
for i, r in enumerate(runs[1:], 1):
    gd = r[0] != ref[0]
    dd = r[1] != ref[1]
    if gd.any() or dd.any():
        wgs = sorted(set(np.nonzero(dd)[0].tolist()))
        print('run %d: gw differs at %d entries (taps %s); checksum slots differ: %s; WGs %s; threads %s' % (
            i, gd.sum(), sorted(set((np.nonzero(gd)[2] * 3 + np.nonzero(gd)[3]).tolist())), sorted(set(np.nonzero(dd)[2].tolist())), wgs[:6],
            sorted(set(np.nonzero(dd)[1].tolist()))[:12]))
        for wg in wgs[:3]:
            t = np.nonzero(dd[wg])[0][0]
            for sl in range(8):
                th = np.nonzero(dd[wg][:, sl])[0].tolist()
                if th:
                    print('   WG %d slot %d: differing threads %s' % (wg, sl, th[:48]))
            print('   WG %d thread %d: ref %s' % (wg, t, ref[1][wg, t]), '\n                     now %s' % r[1][wg, t])
    else:
        print('run %d: identical' % i)
