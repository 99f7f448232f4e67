#!/usr/bin/env python
"""round-5 debug aid: run-to-run reproducibility of the DCN weight gradient (it has no atomics: must be bit-identical)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF

def run(B, C, Co, dg, H, W, ostd, n=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda:0'
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    off = (torch.randn(B, dg * 18, H, W, generator=g) * ostd).to(dev)
    m = torch.rand(B, dg * 9, H, W, generator=g).to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    gout = torch.randn(B, Co, H, W, generator=g).to(dev)
    outs = []
    for i in range(n):
        ls = [t.clone().requires_grad_(True) for t in (x, off, m, w, b)]
        out = RF.modulated_deform_conv(*ls, 1, 1, 1, 1, dg)
        out.backward(gout)
        torch.cuda.synchronize()
        outs.append((ls[3].grad.clone(), ls[4].grad.clone(), ls[1].grad.clone(), ls[2].grad.clone()))
    ref = outs[0]
    bad = []
    for i, o in enumerate(outs[1:], 1):
        if not torch.equal(o[0], ref[0]):
            d = (o[0] - ref[0]).abs()
            taps = [k for k in range(9) if d[:, :, k // 3, k % 3].max() > 0]
            bad.append((i, float(d.max() / ref[0].abs().max()), taps))
    same_rest = all(torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]) and torch.equal(o[3], ref[3]) for o in outs[1:])
    print('B%d C%d Co%d %dx%d ostd %g: gw differs in %d of %d repeats %s; gb/goff/gmask reproducible: %s' % (B, C, Co, H, W, ostd, len(bad), n - 1, bad[:4], same_rest))

def run_pack(B, C, Co, dg, H, W, ostd, n=8, seed=0, act=0):
    g = torch.Generator().manual_seed(seed)
    dev = 'cuda:0'
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    om = torch.randn(B, dg * 27, H, W, generator=g)
    om[:, :dg * 18] *= ostd
    om = om.to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev).requires_grad_(True)
    b = torch.randn(Co, generator=g).to(dev)
    gout = torch.randn(B, Co, H, W, generator=g).to(dev)
    outs = []
    for i in range(n):
        ls = [x.clone().requires_grad_(True), om.clone().requires_grad_(True), w, b.clone().requires_grad_(True)]
        w.grad = None
        out = RF.dcn_pack(*ls, 1, 1, 1, dg, act, 0.1)
        out.backward(gout)
        torch.cuda.synchronize()
        outs.append((w.grad.clone(), ls[3].grad.clone(), ls[1].grad.clone()))
    ref = outs[0]
    bad = []
    for i, o in enumerate(outs[1:], 1):
        if not torch.equal(o[0], ref[0]):
            d = (o[0] - ref[0]).abs()
            taps = [k for k in range(9) if d[:, :, k // 3, k % 3].max() > 0]
            bad.append((i, float(d.max() / ref[0].abs().max()), taps))
    same_rest = all(torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2]) for o in outs[1:])
    print('pack act%d B%d C%d Co%d %dx%d ostd %g: gw differs in %d of %d repeats %s; gb/gom reproducible: %s' % (act, B, C, Co, H, W, ostd, len(bad), n - 1, bad[:4], same_rest))

for spec in sys.argv[1:]:
    a = spec.split(',')
    if a[0] == 'p':
        run_pack(*[int(v) for v in a[1:7]], float(a[7]), act=int(a[8]) if len(a) > 8 else 0)
    else:
        run(*[int(v) for v in a[:6]], float(a[6]))
