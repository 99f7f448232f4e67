"""Parity spot-check run in a SUBPROCESS by tests/test_gpu_switches.py with one developer switch set (the switches are read once
per process): a fused DCN pack forward + backward against the CPU oracle at three offset scales, and a 3x3 / 1x1 / stride-2
conv block against float64 torch.  Exit code 0 = all within the op-level bf16x3 tolerance (1e-4 of the tensor's max)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from conftest import rel_err  # noqa: E402


def main():
    import torch.nn.functional as F
    from oracle.dcn_oracle import modulated_deform_conv as oracle_dcn
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.dcn import modulated_deform_conv
    d = torch.device('cuda:0')
    worst = 0.0
    for C, Co, H, W, ostd in ((64, 64, 20, 40, 0.3), (64, 64, 24, 40, 4.0), (128, 128, 12, 40, 7.0), (32, 72, 9, 33, 1.0)):
        g = torch.Generator().manual_seed(C + H)
        t = [torch.randn(1, C, H, W, generator=g), torch.randn(1, 144, H, W, generator=g) * ostd, torch.rand(1, 72, H, W, generator=g),
             torch.randn(Co, C, 3, 3, generator=g) / (3 * C ** 0.5), torch.randn(Co, generator=g)]
        gout = torch.randn(1, Co, H, W, generator=g)
        ref = [x.clone().requires_grad_(True) for x in t]
        oracle_dcn(*ref, 1, 1, 1, 1, 8).backward(gout)
        got = [x.to(d).requires_grad_(True) for x in t]
        out = modulated_deform_conv(*got, 1, 1, 1, 1, 8)
        out.backward(gout.to(d))
        errs = [rel_err(a.grad.cpu(), b.grad) for a, b in zip(got, ref)]
        errs.append(rel_err(out.detach().cpu(), oracle_dcn(*[x.detach() for x in ref], 1, 1, 1, 1, 8)))
        print('dcn', C, Co, H, W, ostd, ' '.join('%.1e' % e for e in errs))
        worst = max(worst, max(errs))
    # (the 24 x 128 frame takes the 8 x 64 tile of conv_fwd5, the 36 x 72 ones its 16 x 32 tile)
    for k, s, Ci, Co, Hc, Wc in ((3, 1, 64, 64, 36, 72), (3, 1, 64, 64, 24, 128), (3, 1, 128, 216, 36, 72), (1, 1, 64, 64, 36, 72), (3, 2, 64, 64, 36, 72)):
        g = torch.Generator().manual_seed(k * 10 + s)
        conv = torch.nn.Conv2d(Ci, Co, k, s, k // 2)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (3.0 * Ci ** 0.5))
            conv.bias.copy_(torch.randn(Co, generator=g) * 0.1)
        x = torch.randn(2, Ci, Hc, Wc, generator=g)
        wr, xr = conv.weight.detach().double().requires_grad_(True), x.double().requires_grad_(True)
        z = F.conv2d(xr, wr, conv.bias.detach().double(), s, k // 2)
        yr = F.leaky_relu(z, 0.1)
        # (away from the activation kink: where the exact pre-activation is within 1e-3 of zero the derivative legitimately depends on
        # the last bits of the GEMM -- same masking as tests/test_gpu_conv.py)
        gout = torch.randn(yr.shape, generator=g, dtype=torch.float64) * (z.detach().abs() > 1e-3).double()
        yr.backward(gout)
        conv = conv.to(d)
        xg = x.to(d).requires_grad_(True)
        y = RF.conv2d(xg, conv, act=RF.ACT_LRELU, slope=0.1)
        y.backward(gout.float().to(d))
        errs = [rel_err(y.detach().cpu(), yr.detach()), rel_err(xg.grad.cpu(), xr.grad), rel_err(conv.weight.grad.cpu(), wr.grad)]
        print('conv', k, s, Ci, Co, Hc, Wc, ' '.join('%.1e' % e for e in errs))
        worst = max(worst, max(errs))
    # residual block (arch_util.ResidualBlock_noBN) as one autograd node: relu' of the hidden activation is applied in the epilogue of
    # conv2's data gradient on frames the 8 x 64 tile takes (24 x 128), by the consumers otherwise (36 x 72)
    for Hc, Wc in ((24, 128), (36, 72)):
        g = torch.Generator().manual_seed(Hc)
        c1, c2 = torch.nn.Conv2d(64, 64, 3, 1, 1), torch.nn.Conv2d(64, 64, 3, 1, 1)
        with torch.no_grad():
            for c in (c1, c2):
                c.weight.copy_(torch.randn(c.weight.shape, generator=g) / 24.0)
                c.bias.copy_(torch.randn(64, generator=g) * 0.1)
        x = torch.randn(2, 64, Hc, Wc, generator=g)
        xr = x.double().requires_grad_(True)
        w = [t.detach().double().requires_grad_(True) for t in (c1.weight, c1.bias, c2.weight, c2.bias)]
        z = F.conv2d(xr, w[0], w[1], padding=1)
        yr = xr + F.conv2d(F.relu(z), w[2], w[3], padding=1)
        gout = torch.randn(yr.shape, generator=g, dtype=torch.float64)
        yr.backward(gout)
        c1, c2 = c1.to(d), c2.to(d)
        xg = x.to(d).requires_grad_(True)
        y = RF.res_block(xg, c1, c2)
        y.backward(gout.float().to(d))
        # (gradient entries next to a ReLU kink of the hidden activation depend on the last bits of conv1: compared by relative L2 norm)
        l2 = lambda a, b: ((a.double().cpu() - b).norm() / b.norm()).item()   # noqa: E731
        errs = [rel_err(y.detach().cpu(), yr.detach()), l2(xg.grad, xr.grad), l2(c1.weight.grad, w[0].grad), l2(c1.bias.grad, w[1].grad),
                l2(c2.weight.grad, w[2].grad)]
        print('res_block', Hc, Wc, ' '.join('%.1e' % e for e in errs))
        worst = max(worst, errs[0], max(errs[1:]) / 50)   # (L2 tolerance 5e-3 for the gradients, as tests/test_gpu_net.py: a handful of
        # hidden pre-activations within 1e-5 of zero flip their relu' under the bf16x3 forward's 4e-6 error and move single entries)
    torch.cuda.synchronize()
    print('worst rel_err %.3e' % worst)
    return 0 if worst <= 1e-4 else 1


if __name__ == '__main__':
    sys.exit(main())
