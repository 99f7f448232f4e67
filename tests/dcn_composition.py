"""An oracle-independent construction of the modulated DCN for SPATIALLY CONSTANT offsets (test infrastructure).

For offsets that are the same at every output pixel but different per (deformable group g, tap k), and any per-pixel mask, the
reference operator (deform_conv_cuda_kernel.cu:571-633, bilinear :467-497, host deform_conv_cuda.cpp:490-569; 3 x 3, stride 1,
dilation 1) is

    out[b, o] = bias[o] + sum_{g, k} W[o, g-chans, k] . ( mask[b, g*K + k] * sum_{4 corners} w_c * zshift(x[b, g-chans], corner) )

where zshift(x, a, b)[h, w] = x[h + a, w + b] inside the image and 0 outside, and the corner weights come from the constant fractional
part of (k_y - pad + dy, k_x - pad + dx).  The zero-filled shift reproduces the reference's rule set exactly: a sample contributes only
for -1 < y < H, -1 < x < W (kernel.cu:618) and every corner is guarded on its own (:480-491), so positions in (-1, 0) and (H - 1, H)
interpolate against implicit zeros.  Built from slicing + F.conv2d 1 x 1 only -- no shared code with oracle/dcn_oracle -- it pins
  * the offset channel order  g*2K + 2k (dy), + 1 (dx)          (kernel.cu:608-612),
  * the mask channel order    g*K + k                            (kernel.cu:602, 610, 613),
  * the channel -> group map  c // (C / dg)                      (kernel.cu:592),
  * the column order c*K + k against weight.flatten(1)           (kernel.cu:589, 627; cpp:553),
  * bilinear weights and the zero-outside rule,
and, through autograd, every gradient: grad_input, grad_mask, grad_weight, grad_bias directly, grad_offset summed over (b, h, w)
(d out / d dy[g, k] of a constant field is the sum of the per-pixel offset gradients; the fractional parts are differentiable,
the floor is piecewise constant exactly as in kernel.cu:696-767 away from the integer lattice)."""
import torch
import torch.nn.functional as F


def zshift(x, a, b):
    """zshift(x, a, b)[..., h, w] = x[..., h + a, w + b] inside the image, 0 outside (a, b integers)."""
    H, W = x.shape[-2:]
    out = torch.zeros_like(x)
    h0, h1 = max(0, -a), min(H, H - a)
    w0, w1 = max(0, -b), min(W, W - b)
    if h0 < h1 and w0 < w1:
        out[..., h0:h1, w0:w1] = x[..., h0 + a:h1 + a, w0 + b:w1 + b]
    return out


def constant_offset_dcn(x, dyx, mask, weight, bias, pad=1, dg=1):
    """dyx: (dg, K, 2) tensor of (dy, dx) per (group, tap); mask (B, dg*K, H, W); 3 x 3, stride 1, dilation 1, groups 1."""
    B, C, H, W = x.shape
    Co, K, cpg = weight.shape[0], 9, C // dg
    out = torch.zeros(B, Co, H, W, dtype=x.dtype) if bias is None else bias.view(1, Co, 1, 1).expand(B, Co, H, W).clone()
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg]
        for k in range(K):
            i, j = k // 3, k % 3
            ty, tx = (i - pad) + dyx[g, k, 0], (j - pad) + dyx[g, k, 1]
            fy, fx = torch.floor(ty.detach()), torch.floor(tx.detach())
            ly, lx = ty - fy, tx - fx
            a, b_ = int(fy), int(fx)
            s = (1 - ly) * (1 - lx) * zshift(xg, a, b_) + (1 - ly) * lx * zshift(xg, a, b_ + 1) \
                + ly * (1 - lx) * zshift(xg, a + 1, b_) + ly * lx * zshift(xg, a + 1, b_ + 1)
            col = mask[:, g * K + k:g * K + k + 1] * s
            out = out + F.conv2d(col, weight[:, g * cpg:(g + 1) * cpg, i, j][:, :, None, None])
    return out


def offset_field(dyx, B, H, W):
    """(dg, K, 2) constants -> the operator's (B, dg*2K, H, W) offset tensor: channel g*2K + 2k = dy, + 1 = dx."""
    dg, K, _ = dyx.shape
    return dyx.reshape(1, dg * 2 * K, 1, 1).expand(B, dg * 2 * K, H, W).contiguous()


def make_case(B, C, Co, dg, H, W, seed, dtype=torch.float64, kind='mixed'):
    """Per-(group, tap) constants that exercise every branch of the sampling rule: integers, fractions, positions in (-1, 0) and
    (H - 1, H) for border pixels, and (group 0, tap 0) entirely outside the image."""
    g = torch.Generator().manual_seed(seed)
    K = 9
    if kind == 'integer':
        dyx = torch.randint(-3, 4, (dg, K, 2), generator=g).to(dtype)
    else:
        dyx = (torch.rand(dg, K, 2, generator=g, dtype=torch.float64) * 6 - 3)
        dyx = (dyx * 8).round() / 8          # multiples of 1/8: fractional parts exact in f32 too
        frac = dyx - dyx.floor()
        dyx = torch.where(frac == 0, dyx + 0.375, dyx)   # keep the differentiated cases off the integer lattice
        if kind == 'mixed':
            dyx[0, 1] = torch.tensor([2.0, -1.0])        # one integer tap (no gradient check on it)
            dyx[0, 0] = torch.tensor([float(H + 4), 0.25])   # far outside: contributes nothing
            dyx[-1, 8] = torch.tensor([-0.5, 0.5])       # border pixels sample (-1, 0) / (H - 1, H) strips
        dyx = dyx.to(dtype)
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64).to(dtype)
    mask = torch.rand(B, dg * K, H, W, generator=g, dtype=torch.float64).to(dtype)
    w = (torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64) / (C * 9) ** 0.5).to(dtype)
    b = torch.randn(Co, generator=g, dtype=torch.float64).to(dtype)
    gout = torch.randn(B, Co, H, W, generator=g, dtype=torch.float64).to(dtype)
    return x, dyx, mask, w, b, gout


def composition_reference(x, dyx, mask, w, b, gout, dg):
    """Forward + all gradients of the construction (float64 autograd).  Returns out, (gx, goff_sum[dg, K, 2], gmask, gw, gb)."""
    leaves = [t.detach().double().clone().requires_grad_(True) for t in (x, dyx, mask, w, b)]
    out = constant_offset_dcn(*leaves, pad=1, dg=dg)
    out.backward(gout.double())
    return out.detach(), [l.grad for l in leaves]


def offset_grad_sums(goff, dg):
    """(B, dg*2K, H, W) per-pixel offset gradient -> (dg, K, 2) sums over (b, h, w)."""
    return goff.double().sum(dim=(0, 2, 3)).reshape(dg, 9, 2)


def offset_grad_comparable(dyx, pad=1):
    """(dg, K) bool: taps whose offset gradient the construction defines the way the reference does.  The one exception is a tap
    that puts a row / column of pixels EXACTLY on -1 (an integer displacement <= -1 in either direction): there the operator is not
    differentiable, the reference's range test (kernel.cu:747-750, `h_im > -1`) returns the zero sentinel and the construction the
    right derivative.  (Everywhere else on the integer lattice both take the right derivative; at exactly H / W both give 0.)"""
    dg, K, _ = dyx.shape
    ok = torch.ones(dg, K, dtype=torch.bool)
    for k in range(K):
        ty, tx = (k // 3 - pad) + dyx[:, k, 0], (k % 3 - pad) + dyx[:, k, 1]
        ok[:, k] = ~(((ty == ty.floor()) & (ty <= -1)) | ((tx == tx.floor()) & (tx <= -1)))
    return ok
