"""HIP conv blocks vs torch CPU float64 (exact-f32 MFMA => tight tolerance).  -m gpu"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from gpu_util import check, dev, gemm_modes

pytestmark = pytest.mark.gpu
# exact-f32 MFMA: an fmaf chain -> tight; bf16x3 split: ~2^-17 per product
TOLS = {'f32': 2e-5, 'bf16x3': 1e-4, 'bf16x2': 2e-2, 'bf16': 2e-2, 'f16fp8': 2.5e-4}   # (speed modes: tests/test_gpu_modes.py)

CASES = [
    # C1, C2, Co, k, stride, act, residual, pixel_shuffle, B, H, W
    (16, 0, 16, 3, 1, 'lrelu', False, False, 2, 12, 20),
    (64, 0, 64, 3, 1, 'relu', False, False, 1, 45, 80),
    (64, 0, 64, 3, 1, 'none', True, False, 2, 16, 40),
    (64, 64, 64, 3, 1, 'lrelu', False, False, 1, 23, 37),
    (3, 0, 64, 3, 1, 'lrelu', False, False, 2, 16, 24),
    (64, 0, 3, 3, 1, 'none', True, False, 1, 32, 48),
    (64, 0, 216, 3, 1, 'none', False, False, 1, 20, 36),
    (16, 0, 108, 3, 1, 'none', False, False, 2, 7, 9),
    (64, 0, 64, 3, 2, 'lrelu', False, False, 2, 24, 40),
    (16, 0, 16, 3, 2, 'lrelu', False, False, 1, 16, 24),
    (64, 0, 256, 3, 1, 'lrelu', False, True, 1, 12, 20),
    (16, 0, 64, 3, 1, 'lrelu', False, True, 2, 8, 12),
    (320, 0, 64, 1, 1, 'lrelu', False, False, 1, 16, 40),
    (48, 0, 16, 1, 1, 'none', False, False, 2, 12, 20),
    (32, 32, 16, 1, 1, 'lrelu', False, False, 1, 9, 33),
    (64, 64, 80, 1, 1, 'lrelu', False, False, 2, 10, 36),   # concat 1x1 on the GEMM weight-gradient path (HW % 8 == 0)
    (64, 0, 64, 3, 2, 'relu', False, False, 1, 23, 37),      # odd sizes through the zero-insert data gradient
    (64, 0, 64, 3, 2, 'lrelu', False, False, 2, 32, 48),     # stride-2 weight gradient on the bf16x3 direct-load kernel
    (16, 0, 24, 3, 2, 'none', False, False, 1, 18, 32),      # same, odd output height, Co/C not multiples of 32
    (128, 0, 128, 3, 1, 'relu', False, False, 1, 12, 36),
    (64, 0, 64, 3, 1, 'lrelu', True, False, 1, 16, 24),   # act + residual (sAtt_3 pattern)
    (32, 0, 4, 3, 1, 'lrelu', False, False, 2, 10, 72),    # thin layer (Co <= 4) on the vector-ALU weight-gradient kernel: act', ragged tile
    (16, 0, 1, 3, 1, 'none', False, False, 3, 7, 132),      # thin layer, one output channel, three column tiles
]


def _ref(x1, x2, w, b, res, stride, act, ps):
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    y = F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)
    if ps:
        y = F.pixel_shuffle(y, 2)
    if act == 'relu':
        y = F.relu(y)
    elif act == 'lrelu':
        y = F.leaky_relu(y, 0.1)
    if res is not None:
        y = y + res
    return y


gemm_mode = gemm_modes()


@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join(str(v) for v in c))
def test_conv_block_forward_backward(case, gemm_mode):
    from realvsr_amd import functional as RF
    TOL = TOLS[gemm_mode]
    C1, C2, Co, k, stride, act, use_res, ps, B, H, W = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    conv = nn.Conv2d(C1 + C2, Co, k, stride, k // 2)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (3.0 * (C1 + C2) ** 0.5))
        conv.bias.copy_(torch.randn(Co, generator=g) * 0.1)
    x1 = torch.randn(B, C1, H, W, generator=g)
    x2 = torch.randn(B, C2, H, W, generator=g) if C2 else None
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    oshape = (B, Co // 4, 2 * Ho, 2 * Wo) if ps else (B, Co, Ho, Wo)
    res = torch.randn(oshape, generator=g) if use_res else None
    gout = torch.randn(oshape, generator=g)
    if act != 'none':
        # keep the comparison away from the activation kink: where the exact pre-activation is within
        # 1e-3 of zero the derivative legitimately depends on the last bits of the GEMM
        with torch.no_grad():
            z = _ref(x1.double(), None if x2 is None else x2.double(), conv.weight.double(), conv.bias.double(), None,
                     stride, 'none', ps)
            gout = gout * (z.abs() > (1e-3 if gemm_mode in ('f32', 'bf16x3') else 5e-2)).float()   # (speed modes perturb z by ~2e-3 |z|)

    # float64 CPU reference
    r = [t.double().requires_grad_(True) if t is not None else None for t in (x1, x2, res)]
    wr, br = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yr = _ref(r[0], r[1], wr, br, r[2], stride, act, ps)
    yr.backward(gout.double())

    d = dev()
    conv = conv.to(d)
    t = [v.to(d).requires_grad_(True) if v is not None else None for v in (x1, x2, res)]
    code = {'none': RF.ACT_NONE, 'relu': RF.ACT_RELU, 'lrelu': RF.ACT_LRELU}[act]
    y = RF.conv2d(t[0], conv, code, 0.1, x2=t[1], residual=t[2], pixel_shuffle=ps)
    y.backward(gout.to(d))
    torch.cuda.synchronize()
    check('out', y, yr, TOL)
    check('grad_x1', t[0].grad, r[0].grad, TOL)
    if C2:
        check('grad_x2', t[1].grad, r[1].grad, TOL)
    if use_res:
        check('grad_res', t[2].grad, r[2].grad, TOL)
    check('grad_weight', conv.weight.grad, wr.grad, TOL)
    check('grad_bias', conv.bias.grad, br.grad, TOL)


def _random_cases(n, seed):
    """Seeded sweep over the shapes the buffer-addressed staging has to get right: channel counts that leave partial octets /
    chunks, concat inputs on and off the 16 / 64-channel boundaries, widths on and off the vector path, ragged tiles."""
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        k = rnd.choice([3, 3, 3, 1])
        C1 = rnd.choice([3, 8, 9, 16, 24, 40, 64, 72])
        C2 = rnd.choice([0, 0, 0, 16, 64]) if C1 % 8 == 0 else 0
        Co = rnd.choice([1, 3, 4, 12, 32, 64, 80, 216]) if k == 3 else rnd.choice([16, 64])
        ps = k == 3 and Co % 4 == 0 and Co >= 32 and rnd.random() < 0.25
        act = rnd.choice(['none', 'relu', 'lrelu'])
        use_res = (not ps) and rnd.random() < 0.3
        H = rnd.choice([5, 8, 13, 16, 18, 33])
        W = rnd.choice([8, 12, 20, 36, 68, 30])   # 30: off the vector path
        out.append((C1, C2, Co, k, 1, act, use_res, ps, rnd.choice([1, 2, 3]), H, W))
    return out


@pytest.mark.parametrize('case', _random_cases(36, 20260928), ids=lambda c: '-'.join(str(v) for v in c))
def test_conv_random_shapes(case, gemm_mode):
    test_conv_block_forward_backward(case, gemm_mode)


def test_conv_refuses_cpu_tensors():
    from realvsr_amd import functional as RF
    conv = nn.Conv2d(4, 4, 3, 1, 1)
    with pytest.raises(NotImplementedError):
        RF.conv2d(torch.randn(1, 4, 8, 8), conv)


def test_conv_wgrad_is_deterministic():
    from realvsr_amd import functional as RF
    d = dev()
    torch.manual_seed(0)
    conv = nn.Conv2d(64, 64, 3, 1, 1).to(d)
    x = torch.randn(4, 64, 45, 80, device=d)
    grads = []
    for _ in range(2):
        conv.zero_grad()
        RF.conv2d(x, conv, RF.ACT_LRELU).square().sum().backward()
        grads.append(conv.weight.grad.clone())
    assert torch.equal(grads[0], grads[1])
