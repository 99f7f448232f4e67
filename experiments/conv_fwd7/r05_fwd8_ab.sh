#!/bin/bash
# conv_fwd8 (half of the epilogue in the next tile's MFMA shadow) against conv_fwd5 in one call: bit identity, micro A/B, conv tests, step A/B
mkdir -p gpurun_out
{
python - <<'P'
import os, torch, torch.nn as nn
from realvsr_amd import functional as RF
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, C, Co, H, W, act, res) in [(3, 64, 64, 180, 320, RF.ACT_LRELU, False), (2, 64, 64, 180, 320, RF.ACT_NONE, True), (2, 128, 128, 64, 128, RF.ACT_RELU, False),
                                    (1, 48, 64, 37, 64, RF.ACT_LRELU, True), (5, 64, 64, 8, 64, RF.ACT_NONE, False), (2, 48, 40, 100, 192, RF.ACT_LRELU, False), (2, 64, 64, 64, 64, RF.ACT_LRELU, False)]:
    conv = nn.Conv2d(C, Co, 3, 1, 1).to(dev)
    x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    r = torch.randn(B, Co, H, W, device=dev) if res else None
    g = torch.randn(B, Co, H, W, device=dev)
    outs = []
    for sw in ('0', '1'):
        os.environ['RVSR_CONV_FWD8'] = sw
        x.grad = None
        y = RF.conv2d(x, conv, act, residual=r) if res else RF.conv2d(x, conv, act)
        y.backward(g)
        outs.append((y.detach().clone(), x.grad.clone()))
    torch.cuda.synchronize()
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print('fwd8 vs fwd5', (B, C, Co, H, W, act, res), 'bit-identical (output and data gradient)' if same else 'DIFFERENT max %g / %g' % ((outs[0][0] - outs[1][0]).abs().max().item(), (outs[0][1] - outs[1][1]).abs().max().item()), flush=True)
P
for r in 1 2; do
for sw in 0 1; do
  echo -n "FWD8=$sw fwd: "; RVSR_CONV_FWD8=$sw timeout 120 python tools/conv_micro.py --iters 30 2>&1 | tail -1
  echo -n "FWD8=$sw fwd+bwd: "; RVSR_CONV_FWD8=$sw timeout 120 python tools/conv_micro.py --iters 20 --bwd 2>&1 | tail -1
done
done
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
for r in 1 2; do
for sw in 0 1; do
  echo -n "FWD8=$sw step: "; RVSR_CONV_FWD8=$sw timeout 600 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline --no-sweep 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline_conv']['avg_launch_ms'], d['roofline']['frac'])"
done
done
} > gpurun_out/r05_fwd8_ab.log 2>&1
cat gpurun_out/r05_fwd8_ab.log
