import os, sys, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
import realvsr_amd
from realvsr_amd import functional as RF, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
import torch.nn.functional as F
for (B, C, Co, H, W, act, res) in [(2, 64, 64, 180, 320, RF.ACT_LRELU, False), (2, 64, 64, 90, 160, RF.ACT_NONE, True), (1, 128, 128, 64, 128, RF.ACT_RELU, False), (1, 48, 40, 50, 64, RF.ACT_LRELU, False), (2, 64, 32, 64, 64, RF.ACT_NONE, False)]:
    conv = nn.Conv2d(C, Co, 3, 1, 1).to(dev)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    r = torch.randn(B, Co, H, W, device=dev) if res else None
    ref = F.conv2d(x.detach().double(), conv.weight.double(), conv.bias.double(), padding=1)
    if act == RF.ACT_LRELU: ref = F.leaky_relu(ref, 0.1)
    if act == RF.ACT_RELU: ref = F.relu(ref)
    if res: ref = ref + r.double()
    errs = {}
    for mode in ('bf16x3', 'f16fp8'):
        _lib.set_gemm_mode(mode)
        assert _lib.get_gemm_mode() == mode
        x.grad = None
        y = RF.conv2d(x, conv, act, residual=r) if res else RF.conv2d(x, conv, act)
        y.backward(torch.ones_like(y))
        errs[mode] = ((y.detach().double() - ref).norm() / ref.norm()).item()
        errs[mode + '_gx'] = x.grad.clone()
    print((B, C, Co, H, W, act, res), 'l2 vs f64: bf16x3 %.2e  f16fp8 %.2e   data gradients identical: %s' % (errs['bf16x3'], errs['f16fp8'], torch.equal(errs['bf16x3_gx'], errs['f16fp8_gx'])))
_lib.set_gemm_mode('bf16x3')
