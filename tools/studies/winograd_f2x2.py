#!/usr/bin/env python
"""Numerics + cost study (CPU, no GPU code) of Winograd F(2x2, 3x3) for the 64 -> 64 3x3 stride-1 convolutions in the shipped product
arithmetic (review item 5 of round 5): transforms in f32, the 16 position GEMMs (K = 64) as the three-term bf16 split
u_hi*v_hi + u_hi*v_lo + u_lo*v_hi with f32 accumulation, output transform in f32.  Prints the relative l2 error of the output against a
float64 direct convolution next to the direct form's (K = 576, same split), for the operand statistics of the network:
  * weights kaiming(fan_in) * 0.1 (arch_util.initialize_weights) and PyTorch-default;  activations N(0,1), post-ReLU, and post-ReLU with
    log-normal channel scales."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
bf = torch.bfloat16


def q(x):
    return x.to(bf).to(torch.float32)


def split(x):
    h = q(x)
    return h, q(x - h)


def direct_bf16x3(x, w):
    """the shipped arithmetic: K = (tap, c) GEMM, three bf16 products, f32 accumulation (emulated with f32 matmuls of bf16-exact operands)"""
    B, C, H, W = x.shape
    cols = F.unfold(x, 3, padding=1)                      # (B, C*9, HW)
    wm = w.reshape(w.shape[0], -1)
    ch, cl = split(cols)
    wh, wl = split(wm)
    out = wh @ ch + wh @ cl + wl @ ch
    return out.reshape(B, w.shape[0], H, W)


Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_bf16x3(x, w, split_u_in_f64=False):
    B, C, H, W = x.shape
    Co = w.shape[0]
    U = torch.einsum('ai,ocij,bj->abco', G, w, G)                       # (4, 4, C, Co), f32 (computed once per optimizer step)
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # (B, C, H/2, W/2, 4, 4)
    V = torch.einsum('ai,bcyxij,ej->aebcyx', Bt, d, Bt)                 # (4, 4, B, C, ty, tx), f32 adds only (exact up to f32 rounding)
    Uh, Ul = split(U)
    Vh, Vl = split(V)
    M = torch.einsum('abco,abncyx->abnoyx', Uh, Vh) + torch.einsum('abco,abncyx->abnoyx', Uh, Vl) + torch.einsum('abco,abncyx->abnoyx', Ul, Vh)
    Y = torch.einsum('ia,abnoyx,jb->noyixj', At, M, At)                 # (B, Co, ty, 2, tx, 2)
    return Y.reshape(B, Co, H, W)


def run(name, x, w):
    ref = F.conv2d(x.double(), w.double(), padding=1)
    den = ref.norm()
    e = lambda y: ((y.double() - ref).norm() / den).item()   # noqa: E731
    print('%-58s direct f32 %.2e | direct bf16x3 %.2e | Winograd f32 %.2e | Winograd bf16x3 %.2e'
          % (name, e(F.conv2d(x, w, padding=1)), e(direct_bf16x3(x, w)),
             e(torch.einsum('ia,abnoyx,jb->noyixj', At, torch.einsum('abco,abncyx->abnoyx', torch.einsum('ai,ocij,bj->abco', G, w, G),
                            torch.einsum('ai,bcyxij,ej->aebcyx', Bt, F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2), Bt)), At).reshape(ref.shape)),
             e(winograd_bf16x3(x, w))))


C = Co = 64
x0 = torch.randn(2, C, 32, 48)
w_k = torch.randn(Co, C, 3, 3) * (2.0 / (C * 9)) ** 0.5 * 0.1
w_d = (torch.rand(Co, C, 3, 3) * 2 - 1) / (C * 9) ** 0.5
scales = torch.exp(torch.randn(1, C, 1, 1))
run('N(0,1) activations, kaiming*0.1 weights', x0, w_k)
run('N(0,1) activations, default-init weights', x0, w_d)
run('post-ReLU activations, kaiming*0.1 weights', x0.relu(), w_k)
run('post-ReLU, log-normal channel scales, kaiming*0.1', x0.relu() * scales, w_k)
run('gradient-like (N(0,1) * 1e-3), default-init weights', x0 * 1e-3, w_d)

print('''
cost per 2x2 output tile of a 64 -> 64 layer (bf16x3): matrix work 16 x 64 x 64 x 2 x 3 = 393 K bf16-FLOP against 885 K direct (2.25 x fewer);
transformed weights U: 16 positions x 64 x 64 x (hi + lo) x 2 B = 262 KB per layer -- more than the 160 KB of LDS, so U streams through the CU
once per pixel block: a block whose 4 x 32 f32 output accumulators per wave fill the register file (one wave per SIMD, 4 waves x 32 tiles = 512 px)
pays 262 KB / 512 px = 512 B of weight traffic per pixel = ~51 cycles per pixel and CU at the ~10 B/clk a CU fetches, against the 28.5 cycles of
its own matrix + vector work (4 position rows x 96 MFMAs x ~38 cycles per 128 px and SIMD) and the 93 cycles per pixel the shipped direct kernel
takes (MFMA floor 54).''')
