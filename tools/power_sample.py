#!/usr/bin/env python
"""Package power and clock while one kernel runs in a loop (rocm-smi sampled from the host every ~0.3 s, ~4 s per case):
conv_fwd5 (40 x 64 x 180 x 320), conv_wgrad2 (same layer), the DCN forward / backward, and the register-resident MFMA stream of
rvsr_debug_mfma_rate on constant and on N(0,1)-like operands.  Answers: is a kernel that keeps the matrix pipe half busy already at the
package power limit?"""
import os, re, subprocess, sys, threading, time
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realvsr_amd import functional as RF, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
conv = nn.Conv2d(64, 64, 3, 1, 1).to(dev)
x = torch.randn(40, 64, 180, 320, device=dev)
xg = x.clone().requires_grad_(True)
L = _lib.lib()


def mfma(pattern):
    n = 8 * 512 * 8
    ops = torch.ones(n) if pattern == 'ones' else torch.randn(n)
    ops = ops.to(torch.bfloat16).to(dev)
    out = torch.empty(256 * 512, device=dev)
    def run():
        L.rvsr_debug_mfma_rate(ops.data_ptr(), out.data_ptr(), 256, 20000, torch.cuda.current_stream().cuda_stream)
    return run


def fwd():
    with torch.no_grad():
        RF.conv2d(x, conv, RF.ACT_LRELU)


def fwdbwd():
    y = RF.conv2d(xg, conv, RF.ACT_LRELU)
    y.backward(x)


gd = torch.Generator().manual_seed(0)
dx = torch.randn(40, 64, 180, 320, generator=gd).to(dev)
dom = torch.randn(40, 216, 180, 320, generator=gd)
dom[:, :144] *= 1.25
dom = dom.to(dev)
dw = (torch.randn(64, 64, 3, 3, generator=gd) / 24).to(dev)
db = torch.zeros(64, device=dev)
dxg, domg, dwg = dx.clone().requires_grad_(True), dom.clone().requires_grad_(True), dw.clone().requires_grad_(True)


def dcn_fwd():
    with torch.no_grad():
        RF.dcn_pack(dx, dom, dw, db, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)


def dcn_fwdbwd():
    out = RF.dcn_pack(dxg, domg, dwg, db, 1, 1, 1, 8, RF.ACT_LRELU, 0.1)
    out.backward(dx)


cases = [c for c in [('idle', None), ('DCN forward (B = 40, 1.25 px)', dcn_fwd), ('DCN forward + backward', dcn_fwdbwd), ('conv_fwd5 forward', fwd), ('conv fwd + dgrad + wgrad', fwdbwd), ('MFMA stream, operands 1.0', mfma('ones')),
         ('MFMA stream, N(0,1) operands', mfma('normal'))] if not os.environ.get('RVSR_POWER_ONLY') or os.environ['RVSR_POWER_ONLY'] in c[0]]
for name, fn in cases:
    stop = [False]
    count = [0]
    def loop():
        torch.cuda.set_device(0)
        while not stop[0]:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            count[0] += 20
    th = None
    if fn is not None:
        th = threading.Thread(target=loop)
        th.start()
    time.sleep(1.0)
    pw, ck = [], []
    t0, c0 = time.time(), count[0]
    for _ in range(8):
        o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
        m = re.search(r'Power \(W\): ([\d.]+)', o)
        c = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
        if m: pw.append(float(m.group(1)))
        if c: ck.append(int(c.group(1)))
        time.sleep(0.3)
    t1, c1 = time.time(), count[0]
    stop[0] = True
    if th: th.join()
    rate = (c1 - c0) / (t1 - t0) if fn else 0
    print('%-32s power %s W (mean %.0f), sclk %s MHz, %.0f launches/s' % (name, ' '.join('%.0f' % p for p in pw), sum(pw) / max(len(pw), 1),
                                                                       ' '.join(str(c) for c in ck[:4]), rate), flush=True)
