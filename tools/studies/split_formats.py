#!/usr/bin/env python
"""Numerics study (CPU, no GPU code): how many matrix-pipe passes does an f32-grade product need on gfx950?

The shipped GEMMs form a*b as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with bf16 parts (three v_mfma_f32_32x32x16_bf16, ~2^-17 per product).
With the package power limit holding the matrix pipe at ~0.67 of its nominal rate (profiles/r05_mfma_power_micro.txt), the number of
passes per product is what sets the speed of 70 % of a training step.  Candidates with fewer bf16-pass equivalents:
  f16 main + fp8 cross:  a1*b1 in f16 (one pass) + [a1 | a2] x [b2 ; b1] in fp8 e4m3 (K doubled at twice the rate = one pass): 2 passes
where a1 = f16(a), a2 = a - a1 (~2^-12 |a|).  The cross terms need ~5 bits; e4m3 has 4 (implicit bit included) and a narrow exponent
range, so the operands need scaling: per tensor, or per block of 32 along K (the MX scales v_mfma_scale_f32_32x32x64_f8f6f4 applies).
Prints the relative l2 error of C = A @ B (K = 576 = 9 taps x 64 channels) against float64 for each format."""
import torch

torch.manual_seed(0)
M, K, N = 256, 576, 512


def q(x, dt):
    return x.to(dt).to(torch.float32)


def fp8_block(x, dim, block=32):
    """e4m3 with a power-of-two scale per block of `block` elements along `dim` (MX style)."""
    xs = x.movedim(dim, -1)
    shp = xs.shape
    xb = xs.reshape(*shp[:-1], shp[-1] // block, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
    y = (xb * scale).to(torch.float8_e4m3fn).to(torch.float32) / scale
    return y.reshape(shp).movedim(-1, dim)


def fp8_tensor(x):
    scale = 2.0 ** torch.floor(torch.log2(448.0 / x.abs().max()))
    return (x * scale).to(torch.float8_e4m3fn).to(torch.float32) / scale


def study(name, A, B):
    ref = A.double() @ B.double()
    den = ref.norm()

    def err(C):
        return ((C.double() - ref).norm() / den).item()
    out = {}
    ah, bh = q(A, torch.bfloat16), q(B, torch.bfloat16)
    al, bl = q(A - ah, torch.bfloat16), q(B - bh, torch.bfloat16)
    out['f32 (fmaf chain)'] = err(A @ B)
    out['bf16x3 (shipped)'] = err(ah @ bh + ah @ bl + al @ bh)
    out['bf16x2 (speed mode)'] = err(ah @ bh + al @ bh)
    out['bf16 (speed mode)'] = err(ah @ bh)
    a1, b1 = q(A, torch.float16), q(B, torch.float16)
    a2, b2 = A - a1, B - b1
    out['f16 main only'] = err(a1 @ b1)
    out['f16x3'] = err(a1 @ b1 + a1 @ q(b2, torch.float16) + q(a2, torch.float16) @ b1)
    out['f16 main + fp8 cross, per-tensor scale'] = err(a1 @ b1 + fp8_tensor(a1) @ fp8_tensor(b2) + fp8_tensor(a2) @ fp8_tensor(b1))
    out['f16 main + fp8 cross, scale per 32 along K'] = err(a1 @ b1 + fp8_block(a1, 1) @ fp8_block(b2, 0) + fp8_block(a2, 1) @ fp8_block(b1, 0))
    out['f16 main + bf16 cross (3 passes, reference point)'] = err(a1 @ b1 + q(a1, torch.bfloat16) @ q(b2, torch.bfloat16) + q(a2, torch.bfloat16) @ q(b1, torch.bfloat16))
    print(name)
    for k, v in out.items():
        print('   %-52s %.2e' % (k, v))


A = torch.randn(M, K)
B = torch.randn(K, N) / 24
study('N(0,1) activations x N(0, 1/24^2) weights', A, B)
A2 = torch.relu(torch.randn(M, K)) * torch.exp(torch.randn(1, K))          # post-ReLU, per-channel scales over ~e^+-2
study('post-ReLU activations with log-normal channel scales', A2, B)
G = torch.randn(M, K) * torch.exp(2.0 * torch.randn(M, 1)) * 1e-4             # gradients: small, rows over orders of magnitude
study('gradient-like left operand (1e-4, log-normal row scales)', G, B)
