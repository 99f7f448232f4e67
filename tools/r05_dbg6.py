#!/usr/bin/env python
"""round-5 debug aid: the fused DCN backward (dcn6) against the CPU oracle, per-channel / per-tap error maps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF
from oracle import dcn_oracle as O

def run(B, C, Co, dg, H, W, ostd, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, dg * 18, H, W, generator=g) * ostd
    m = torch.rand(B, dg * 9, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)
    b = torch.randn(Co, generator=g)
    gout = torch.randn(B, Co, H, W, generator=g)
    def go(fn, dev):
        ls = [t.clone().to(dev).requires_grad_(True) for t in (x, off, m, w, b)]
        out = fn(*ls, 1, 1, 1, 1, dg)
        out.backward(gout.to(dev))
        return [out.detach().cpu()] + [t.grad.cpu() for t in ls]
    ref = go(O.modulated_deform_conv, 'cpu')
    got = go(RF.modulated_deform_conv, 'cuda:0')
    names = ('out', 'gx', 'goff', 'gmask', 'gw', 'gb')
    print('--- B%d C%d Co%d dg%d %dx%d ostd %g' % (B, C, Co, dg, H, W, ostd))
    for n, a, r in zip(names, got, ref):
        print('%-6s rel %.3e' % (n, ((a - r).abs().max() / r.abs().max()).item()))
    a, r = got[1], ref[1]
    print('gx per channel:', ['%.1e' % ((a[:, c] - r[:, c]).abs().max() / r.abs().max()).item() for c in range(min(C, 16))])
    a, r = got[2], ref[2]
    print('goff per plane (group 0):', ['%.1e' % ((a[:, c] - r[:, c]).abs().max() / r.abs().max()).item() for c in range(18)])
    a, r = got[3], ref[3]
    print('gmask per plane (group 0):', ['%.1e' % ((a[:, c] - r[:, c]).abs().max() / r.abs().max()).item() for c in range(9)])
    a, r = got[4], ref[4]
    print('gw per tap:', ['%.1e' % ((a[:, :, k // 3, k % 3] - r[:, :, k // 3, k % 3]).abs().max() / r.abs().max()).item() for k in range(9)])
    print('gw per channel:', ['%.0e' % ((a[:, c] - r[:, c]).abs().max() / r.abs().max()).item() for c in range(C)])
    print('gw per o (first 8, 32..39):', ['%.1e' % ((a[o] - r[o]).abs().max() / r.abs().max()).item() for o in list(range(8)) + list(range(32, 40))])
    a, r = got[1], ref[1]
    e = (a - r).abs().amax(dim=(0, 1))
    print('gx err rows:', ['%.0e' % v for v in e.amax(dim=1).tolist()])

import sys
if len(sys.argv) > 1:
    for spec in sys.argv[1:]:
        a = spec.split(',')
        run(*[int(v) for v in a[:6]], float(a[6]))
else:
    run(1, 64, 64, 8, 8, 32, 0.0)
    run(1, 64, 64, 8, 8, 32, 0.3)
    run(2, 64, 64, 8, 20, 36, 2.0)
