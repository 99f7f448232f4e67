"""Run-to-run bit-identity of the DCN weight / bias / offset / mask gradients (they have no atomics: dcn_bwdw6's partials are reduced in a
fixed order; only grad_input is an atomic scatter, as in the reference).  Imported by tests/test_gpu_dcn.py and run as a script in a
subprocess for developer switches that are read once per process (RVSR_BWDW6_WG=2).  Exit code 0 = every repeat bit-identical."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# (B, C, Co, dg, H, W, offset std in px): nf64 and nf128 packs, offsets that stay inside the LDS window (0.1 px) and offsets that send
# lanes through the far path and the large windows (6 px); ragged tiles in both directions
CASES = [(3, 64, 64, 8, 45, 80, 0.1), (3, 64, 64, 8, 45, 80, 6.0), (2, 128, 128, 8, 36, 72, 0.1), (2, 128, 128, 8, 36, 72, 6.0),
         (8, 64, 64, 8, 90, 160, 1.25)]


def run_case(case, repeats=20, seed=0):
    from realvsr_amd import functional as RF
    B, C, Co, dg, H, W, ostd = case
    g = torch.Generator().manual_seed(seed)
    dev = torch.device('cuda:0')
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    off = (torch.randn(B, dg * 18, H, W, generator=g) * ostd).to(dev)
    m = torch.rand(B, dg * 9, H, W, generator=g).to(dev)
    w = (torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(dev)
    b = torch.randn(Co, generator=g).to(dev)
    gout = torch.randn(B, Co, H, W, generator=g).to(dev)
    ref, bad = None, []
    for i in range(repeats):
        ls = [t.clone().requires_grad_(True) for t in (x, off, m, w, b)]
        RF.modulated_deform_conv(*ls, 1, 1, 1, 1, dg).backward(gout)
        torch.cuda.synchronize()
        got = {'grad_weight': ls[3].grad, 'grad_bias': ls[4].grad, 'grad_offset': ls[1].grad, 'grad_mask': ls[2].grad}
        if ref is None:
            ref = {k: v.clone() for k, v in got.items()}
            continue
        for k, v in got.items():
            if not torch.equal(v, ref[k]):
                bad.append((i, k, float((v - ref[k]).abs().max() / ref[k].abs().max())))
    return bad


def main():
    failed = 0
    for case in CASES:
        bad = run_case(case)
        print('B%d C%d Co%d dg%d %dx%d ostd %g: %s' % (*case, 'bit-identical' if not bad else 'DIFFERS %s' % bad[:6]))
        failed += bool(bad)
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
