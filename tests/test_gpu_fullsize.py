"""Size-independent properties of the hot-path kernels at BASELINE config 2's full sizes (40 frames of
64 x 180 x 320 in the per-frame / alignment stage, 8 windows elsewhere), where the CPU oracle would take minutes:
translation equivariance (bit-exact: exercises every tile seam), linearity, batch additivity of the weight
gradient, and the DCN <-> plain-conv identity between two independently written kernels.  -m gpu"""
import pytest
import torch
import torch.nn as nn

from gpu_util import check, dev, gemm_modes

gemm_mode = gemm_modes()
pytestmark = pytest.mark.gpu
TOL = {'f32': 2e-5, 'bf16x3': 1e-4}


def _conv(cin=64, cout=64, seed=3):
    torch.manual_seed(seed)
    c = nn.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        c.weight.mul_(0.5)
    return c.to(dev())


def test_conv_translation_equivariance_is_bit_exact(gemm_mode):
    """Shifting the input by (3 rows, 5 columns) shifts the output identically away from the borders: every output
    pixel is the same sum in the same order whichever tile / wave / lane computes it."""
    from realvsr_amd import functional as RF
    conv = _conv()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    xs = torch.zeros_like(x)
    xs[:, :, 3:, 5:] = x[:, :, :-3, :-5]
    with torch.no_grad():
        y, ys = RF.conv2d(x, conv, RF.ACT_LRELU), RF.conv2d(xs, conv, RF.ACT_LRELU)
    # ys(r, c) == y(r - 3, c - 5) wherever both 3x3 footprints see the same data (not the last row / column of ys)
    assert torch.equal(ys[:, :, 5:-1, 7:-1], y[:, :, 2:-4, 2:-6])


def test_conv_linearity_and_wgrad_batch_additivity(gemm_mode):
    from realvsr_amd import functional as RF
    conv = _conv()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    b = torch.randn(40, 64, 180, 320, generator=g).to(dev())
    with torch.no_grad():
        ya, yb, yab = RF.conv2d(a, conv), RF.conv2d(b, conv), RF.conv2d(2.0 * a - b, conv)
        bias = conv.bias.view(1, -1, 1, 1)
    check('conv(2a - b) == 2 conv(a) - conv(b)', yab - bias, 2.0 * (ya - bias) - (yb - bias), 5 * TOL[gemm_mode])
    # weight gradient of the whole batch == sum of the two halves
    gout = torch.randn(40, 64, 180, 320, generator=g).to(dev())

    def wgrad(x, go):
        conv.zero_grad()
        RF.conv2d(x, conv, RF.ACT_LRELU).backward(go)
        return conv.weight.grad.clone(), conv.bias.grad.clone()
    gw, gb = wgrad(a, gout)
    gw1, gb1 = wgrad(a[:20].contiguous(), gout[:20].contiguous())
    gw2, gb2 = wgrad(a[20:].contiguous(), gout[20:].contiguous())
    check('gW(batch) == gW(half 1) + gW(half 2)', gw, gw1 + gw2, 5 * TOL[gemm_mode])
    check('gb(batch) == gb(half 1) + gb(half 2)', gb, gb1 + gb2, 5 * TOL[gemm_mode])


def test_dcn_with_zero_offsets_is_the_plain_conv(gemm_mode):
    """Two independently written kernels (fused DCN vs conv block) agree at the alignment-stage size, forward and
    all three gradients that have a plain-conv counterpart."""
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.dcn import modulated_deform_conv
    conv = _conv(seed=5)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(8, 64, 180, 320, generator=g).to(dev())
    gout = torch.randn(8, 64, 180, 320, generator=g).to(dev())
    off = torch.zeros(8, 8 * 18, 180, 320, device=dev(), requires_grad=True)
    msk = torch.ones(8, 8 * 9, 180, 320, device=dev(), requires_grad=True)
    xd = x.clone().requires_grad_(True)
    wd, bd = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    yd = modulated_deform_conv(xd, off, msk, wd, bd, 1, 1, 1, 1, 8)
    yd.backward(gout)
    xc = x.clone().requires_grad_(True)
    conv.zero_grad()
    yc = RF.conv2d(xc, conv)
    yc.backward(gout)
    tol = 5 * TOL[gemm_mode]
    check('forward', yd, yc, tol)
    check('grad_input', xd.grad, xc.grad, tol)
    check('grad_weight', wd.grad, conv.weight.grad, tol)
    check('grad_bias', bd.grad, conv.bias.grad, tol)
    # mask gradient of a unit mask = per-tap contribution; summed over taps and groups it is <gout, y - bias>
    lhs = float(msk.grad.double().sum())
    rhs = float((gout.double() * (yc.detach().double() - conv.bias.detach().view(1, -1, 1, 1).double())).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(rhs) + 1e-3, (lhs, rhs)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE config 3 (nf128, 7 frames, batch 16: 112 frames of 128 x 180 x 320 in the per-frame / alignment stage)
# and config 5 (nf128, 7 frames, 540 x 960 LR, forward only): the same properties at those sizes.  These are the
# shapes that take the wide paths: two m-blocks / MT = 4 accumulators, 16 channels per deformable group, the two-pass
# (Co > 64) DCN backward, 8 K-chunks per tile.
def _conv128(seed=13):
    torch.manual_seed(seed)
    c = nn.Conv2d(128, 128, 3, 1, 1)
    with torch.no_grad():
        c.weight.mul_(0.5)
    return c.to(dev())


def test_config3_conv_equivariance_linearity_and_wgrad_additivity(gemm_mode):
    from realvsr_amd import functional as RF
    conv = _conv128()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(112, 128, 180, 320, generator=g).to(dev())
    xs = torch.zeros_like(x)
    xs[:, :, 3:, 5:] = x[:, :, :-3, :-5]
    with torch.no_grad():
        y, ys = RF.conv2d(x, conv, RF.ACT_LRELU), RF.conv2d(xs, conv, RF.ACT_LRELU)
    assert torch.equal(ys[:, :, 5:-1, 7:-1], y[:, :, 2:-4, 2:-6])          # bit-exact at every tile seam
    del xs, ys
    with torch.no_grad():
        y2 = RF.conv2d(x * 2.0, conv)
        bias = conv.bias.view(1, -1, 1, 1)
        y1 = RF.conv2d(x, conv)
    check('conv(2x) == 2 conv(x)', y2 - bias, 2.0 * (y1 - bias), 5 * TOL[gemm_mode])
    del y, y1, y2
    gout = torch.randn(112, 128, 180, 320, generator=g).to(dev())

    def wgrad(xx, go):
        conv.zero_grad()
        RF.conv2d(xx, conv, RF.ACT_LRELU).backward(go)
        return conv.weight.grad.clone(), conv.bias.grad.clone()
    gw, gb = wgrad(x, gout)
    gw1, gb1 = wgrad(x[:48].contiguous(), gout[:48].contiguous())
    gw2, gb2 = wgrad(x[48:].contiguous(), gout[48:].contiguous())
    check('gW(112 frames) == gW(48) + gW(64)', gw, gw1 + gw2, 5 * TOL[gemm_mode])
    check('gb(112 frames) == gb(48) + gb(64)', gb, gb1 + gb2, 5 * TOL[gemm_mode])


def test_config3_dcn_zero_offsets_is_the_plain_conv_including_two_pass_backward(gemm_mode):
    """C = Co = 128, 8 deformable groups of 16 channels, one batch-16 slice of the 112-frame alignment call, 180 x 320:
    fused DCN forward and the Co > 64 two-pass backward (grad_input, grad_weight, grad_bias, mask-gradient identity)
    against the independently written conv block."""
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.dcn import modulated_deform_conv
    conv = _conv128(seed=15)
    g = torch.Generator().manual_seed(22)
    B = 16
    x = torch.randn(B, 128, 180, 320, generator=g).to(dev())
    gout = torch.randn(B, 128, 180, 320, generator=g).to(dev())
    off = torch.zeros(B, 8 * 18, 180, 320, device=dev(), requires_grad=True)
    msk = torch.ones(B, 8 * 9, 180, 320, device=dev(), requires_grad=True)
    xd = x.clone().requires_grad_(True)
    wd, bd = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    yd = modulated_deform_conv(xd, off, msk, wd, bd, 1, 1, 1, 1, 8)
    yd.backward(gout)
    xc = x.clone().requires_grad_(True)
    conv.zero_grad()
    yc = RF.conv2d(xc, conv)
    yc.backward(gout)
    tol = 5 * TOL[gemm_mode]
    check('forward', yd, yc, tol)
    check('grad_input (two passes of 64 output channels)', xd.grad, xc.grad, tol)
    check('grad_weight', wd.grad, conv.weight.grad, tol)
    check('grad_bias', bd.grad, conv.bias.grad, tol)
    lhs = float(msk.grad.double().sum())
    rhs = float((gout.double() * (yc.detach().double() - conv.bias.detach().view(1, -1, 1, 1).double())).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(rhs) + 1e-3, (lhs, rhs)
    # integer offsets = shifted sampling: offset (dy, dx) = (2, -3) on every tap of every group equals the plain conv of
    # the input shifted by (-2, +3), away from the border (there the conv pads with zeros where the sampler still
    # reads real pixels of the unshifted image)
    with torch.no_grad():
        off2 = torch.zeros_like(off)
        off2[:, 0::2] = 2.0
        off2[:, 1::2] = -3.0
        ysh = modulated_deform_conv(x, off2, msk.detach(), wd.detach(), bd.detach(), 1, 1, 1, 1, 8)
        xsh = torch.zeros_like(x)
        xsh[:, :, :-2, 3:] = x[:, :, 2:, :-3]
        yref = RF.conv2d(xsh, conv)
    check('integer offsets == conv of the shifted input', ysh[:, :, 3:-3, 4:-4], yref[:, :, 3:-3, 4:-4], tol)


def test_config5_forward_properties_and_graph_runner():
    """540 x 960 LR, nf128, 7 frames (BASELINE config 5), forward only:
      * the nf128 conv block is bit-exactly translation-equivariant at this frame size (7 frames, every tile seam);
      * SlidingWindowRunner(use_graph=True) on a 7-frame 540 x 960 clip: every output frame equals the plain
        single-window forward of the net on that frame's window (bit for bit), output 2160 x 3840."""
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.infer import SlidingWindowRunner, index_generation
    from weights import fill_state_dict
    conv = _conv128(seed=17)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(7, 128, 540, 960, generator=g).to(dev())
    xs = torch.zeros_like(x)
    xs[:, :, 3:, 5:] = x[:, :, :-3, :-5]
    with torch.no_grad():
        y, ys = RF.conv2d(x, conv, RF.ACT_LRELU), RF.conv2d(xs, conv, RF.ACT_LRELU)
    assert torch.equal(ys[:, :, 5:-1, 7:-1], y[:, :, 2:-4, 2:-6])
    del x, xs, y, ys
    net = EDVR(nf=128, nc=3, nframes=7, groups=8, front_RBs=5, back_RBs=2, w_TSA=True)   # back_RBs 2: test time only
    fill_state_dict(net, 303, offset_std=0.03)
    net = net.to(dev()).eval()
    clip = torch.rand(7, 3, 540, 960, generator=g).to(dev())
    run = SlidingWindowRunner(net, 7, padding='reflection', chunk=1, use_graph=True)
    out = run(clip)
    assert out.shape == (7, 3, 2160, 3840) and torch.isfinite(out).all()
    with torch.no_grad():
        for t in (0, 3, 6):            # first / centre / last frame: padded and unpadded windows
            idx = index_generation(t, 7, 7, padding='reflection')
            ref = net(clip[idx].unsqueeze(0))[0]
            assert torch.equal(out[t], ref), t


def test_config5_window_vs_oracle():
    """BASELINE config 5's window against the CPU oracle (EDVR_arch.py:258-320, utils/util.py:222-237): nf128, ONE
    7 x 540 x 960 window, forward only (back_RBs = 2: oracle time), offsets rescaled to a mean of 1 px -- output and PSNR-Y of the plain
    forward AND of the same window through SlidingWindowRunner(use_graph=True) against oracle/edvr_oracle.py."""
    import os
    import bench
    from oracle import edvr_oracle as O
    from realvsr_amd import _lib
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.infer import SlidingWindowRunner
    old = _lib.get_gemm_mode()
    _lib.set_gemm_mode('bf16x3')
    try:
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        torch.manual_seed(0)
        N, H, W = 7, 540, 960
        net = EDVR(nf=128, nc=3, nframes=N, groups=8, front_RBs=5, back_RBs=2, w_TSA=True)
        bench.init_weights(net)
        net = net.to(dev()).eval()
        clip = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(1234))
        xw = clip.unsqueeze(0).to(dev())
        bench.offset_stats(net, xw, 1.0)
        with torch.no_grad():
            out = net(xw)
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            out_o = O.edvr_forward(sd, clip.unsqueeze(0), nframes=N, groups=8, front_RBs=5, back_RBs=2, w_TSA=True)
        assert out.shape == (1, 3, 4 * H, 4 * W)
        check('config-5 window, plain forward', out, out_o, 1e-3)
        # the centre frame of a 7-frame clip IS this window (no temporal padding): the graph-captured runner must reproduce it
        run = SlidingWindowRunner(net, N, padding='replicate', chunk=1, use_graph=True)
        out_g = run(clip.to(dev()))[N // 2:N // 2 + 1]
        assert torch.equal(out_g, out), 'hipGraph sliding-window runner differs from the plain forward on the same window'
        gt = torch.rand(1, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235))

        def psnr_y(o):
            q = (o[:, 0].clamp(0, 1) * 255).round()
            r = (gt.clamp(0, 1) * 255).round()
            return 20 * torch.log10(255.0 / torch.sqrt(((q - r) ** 2).mean()))
        d = abs(psnr_y(out.cpu()).item() - psnr_y(out_o).item())
        print('|PSNR-Y(build, GT) - PSNR-Y(oracle, GT)| = %.2e dB; max abs diff %.2e' % (d, (out.cpu() - out_o).abs().max().item()))
        assert d <= 1e-3
    finally:
        _lib.set_gemm_mode(old)


# ------------------------------------------------------------------------------------------------------------------
# Whole-network oracle parity at BASELINE's own shapes: one seeded window through oracle/edvr_oracle.py
# (torch CPU ops + the OpenMP C DCN restatement) and through the HIP model -- output, loss and EVERY parameter gradient.
def _capture_hip_offsets(net):
    """Forward pre-hooks on every DCN pack: the raw conv_offset_mask output (N x 27 dg x h x w, frame-major) the fused pack consumes --
    the same deterministic conv kernel on the same input, so the same bits."""
    from realvsr_amd import functional as RF
    from realvsr_amd.archs.dcn import ModulatedDeformConvPack
    got, hooks = {}, []

    def make(name, pack):
        def pre(_mod, inputs):
            feat = inputs[0][1] if pack.extra_offset_mask else inputs[0]
            with torch.no_grad():
                got[name] = RF.conv2d(feat, pack.conv_offset_mask).cpu()
        return pre
    for name, m in net.named_modules():
        if isinstance(m, ModulatedDeformConvPack):
            hooks.append(m.register_forward_pre_hook(make(name, m)))
    return got, hooks


def _floor_flips(om_h, om_o, dg):
    """Bilinear samples whose floor() differs between the two implementations' offsets: the sample position is formed exactly as the
    kernels form it (kernel.cu:594-616: `h_in + i * dilation + offset` in f32; stride 1, pad 1, dilation 1), per (frame, group, tap, pixel)."""
    n, _, h, w = om_h.shape
    K = 9
    oy = torch.arange(h, dtype=torch.float32).view(1, 1, 1, h, 1) - 1.0
    ox = torch.arange(w, dtype=torch.float32).view(1, 1, 1, 1, w) - 1.0
    ky = (torch.arange(K) // 3).to(torch.float32).view(1, 1, K, 1, 1)
    kx = (torch.arange(K) % 3).to(torch.float32).view(1, 1, K, 1, 1)
    a = om_h[:, :2 * K * dg].reshape(n, dg, K, 2, h, w)     # channel g * 18 + 2 k + {0: dy, 1: dx}
    b = om_o[:, :2 * K * dg].reshape(n, dg, K, 2, h, w)
    fy = torch.floor((oy + ky) + a[:, :, :, 0]) != torch.floor((oy + ky) + b[:, :, :, 0])
    fx = torch.floor((ox + kx) + a[:, :, :, 1]) != torch.floor((ox + kx) + b[:, :, :, 1])
    return int((fy | fx).sum()), n * dg * K * h * w


def _window_vs_oracle(nf, N, H, W, mode, offset_px, back_rbs=10, explain_flips=True):
    import os
    from oracle import edvr_oracle as O
    from realvsr_amd import loss as L
    from realvsr_amd.archs.EDVR_arch import EDVR
    from gpu_util import l2_err
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    torch.manual_seed(0)
    net = EDVR(nf=nf, nc=3, nframes=N, groups=8, front_RBs=5, back_RBs=back_rbs, w_TSA=True)
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'conv_offset_mask.weight' in name:
                # std chosen so that the mean |offset| is O(offset_px) at the L1 pack: the offset convs see O(1) features
                p.copy_(torch.randn(p.shape, generator=gen) * 0.01)
    x = torch.rand(1, N, 3, H, W, generator=torch.Generator().manual_seed(1234))
    gt = torch.rand(1, 3, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235))
    net = net.to(dev())
    if offset_px:
        import bench
        bench.offset_stats(net, x.to(dev()), offset_px)

    om_o = {}

    def oracle_run(hook):
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        O.om_hook = hook
        try:
            out_o = O.edvr_forward(sd, x, nframes=N, groups=8, front_RBs=5, back_RBs=back_rbs, w_TSA=True)
        finally:
            O.om_hook = None
        loss_o = O.lap_pyr_loss(out_o[:, 0:1], gt[:, 0:1], 3, lf_mode='cb') + O.gw_loss(out_o[:, 1:3], gt[:, 1:3], 4)
        loss_o.backward()
        return sd, out_o, loss_o

    def record(prefix, raw):          # the oracle calls the packs frame by frame (B = 1): call j of a pack = frame j
        om_o.setdefault(prefix, []).append(raw.detach().clone())
        return raw
    sd, out_o, loss_o = oracle_run(record)
    om_h, hooks = _capture_hip_offsets(net)
    out = net(x.to(dev()))
    for hk in hooks:
        hk.remove()
    g = gt.to(dev())
    loss = L.LapPyrLoss(3, 'cb', 'cb', 'mean')(out[:, 0:1], g[:, 0:1]) + L.GWLoss(w=4, reduction='mean')(out[:, 1:3], g[:, 1:3])
    loss.backward()
    # Gradient tolerances at this size: the window draws ~5e7 bilinear samples at ~1 px offsets; an offset that differs in its last
    # bits between the two implementations (different summation order of the offset convs) flips floor() for a handful of them,
    # and each flip changes that sample's offset gradient by O(1).  The sums over all pixels that form the gradients of the offset
    # convs (their biases above all) therefore agree to ~1e-3 .. 1e-2 per tensor, not to the 1e-3 / 5e-3 of the 64 x 64 fixtures;
    # the gradient of the whole parameter vector is bounded separately and much tighter.  That reading is CHECKED below:
    # the flipped samples are counted, and a second oracle run that takes the HIP path's offsets (same floor() everywhere) must agree
    # to the tight bound in every tensor.
    tol_out, tol_loss, tol_g, tol_all = {'f32': (5e-5, 1e-5, 1e-2, 1e-3), 'bf16x3': (1e-3, 1e-4, 2e-2, 2e-3)}[mode]
    check('out', out, out_o.detach(), tol_out)
    assert abs(loss.item() - loss_o.item()) <= tol_loss * abs(loss_o.item()), (loss.item(), loss_o.item())

    def grad_errs(sd_ref):
        worst, name, num, den = 0.0, None, 0.0, 0.0
        for k, p in net.named_parameters():
            e = l2_err(p.grad, sd_ref[k].grad)
            num += float((p.grad.detach().double().cpu() - sd_ref[k].grad.double()).pow(2).sum())
            den += float(sd_ref[k].grad.double().pow(2).sum())
            if e > worst:
                worst, name = e, k
        return (num / den) ** 0.5, worst, name
    e_all, worst, name = grad_errs(sd)
    print('parameter gradients: all %.3e (tol %.1e), worst tensor %.3e (%s, tol %.1e)' % (e_all, tol_all, worst, name, tol_g))
    assert e_all <= tol_all, e_all
    assert worst <= tol_g, (name, worst)
    # PSNR-Y against the synthetic GT, build vs oracle (north_star: within 1e-3 dB)
    def psnr_y(o):
        q = (o[:, 0].clamp(0, 1) * 255).round()
        r = (gt[:, 0].clamp(0, 1) * 255).round()
        return 20 * torch.log10(255.0 / torch.sqrt(((q - r) ** 2).mean()))
    d_psnr = abs(psnr_y(out.detach().cpu()).item() - psnr_y(out_o.detach()).item())
    print('|PSNR-Y(build, GT) - PSNR-Y(oracle, GT)| = %.2e dB' % d_psnr)
    assert d_psnr <= 1e-3
    if not explain_flips:
        return
    # ---- the floor() flips, counted, and the residual without them
    flips = total = 0
    for prefix, frames in om_o.items():
        f, t = _floor_flips(om_h[prefix], torch.cat(frames, 0), 8)
        flips, total = flips + f, total + t
    print('bilinear samples whose floor() differs between the two implementations: %d of %d (%.2e)' % (flips, total, flips / total))
    calls = {}

    def inject(prefix, raw):          # value := the HIP path's offsets / mask logits, gradient := the oracle's own (straight-through)
        j = calls.get(prefix, 0)
        calls[prefix] = j + 1
        return raw + (om_h[prefix][j:j + 1] - raw).detach()
    sd2, out_o2, _ = oracle_run(inject)
    e_all2, worst2, name2 = grad_errs(sd2)
    # Measured (MI355X, round 5): config 2 f32 mode 74 flips of 4.8e7 samples, worst tensor 2.5e-3 -> 2.1e-3 without them; bf16x3 721 flips,
    # 8.3e-3 -> 4.5e-3; config 3 894 of 6.7e7, 5.8e-3 -> 4.4e-3, with the whole parameter vector at 6e-6 .. 4e-5 throughout.  The flips
    # explain about half of the worst tensors' residual; what remains sits in the same few tensors (first offset convs of a PCD level,
    # tAtt_2.bias) whose gradients are sums over ~5e4 pixels with heavy cancellation, so a 1e-5 error of the summands shows as 2..5e-3 of
    # the (small) tensor norm.  Bounds: 2x the measurement, against the 1e-2 / 2e-2 the unmatched comparison needs.
    tol_g2, tol_all2 = {'f32': (4e-3, 2e-4), 'bf16x3': (9e-3, 5e-4)}[mode]
    print('same floor() in both (oracle on the HIP offsets): all %.3e (tol %.1e), worst tensor %.3e (%s, tol %.1e); was %.3e / %.3e with %d flips'
          % (e_all2, tol_all2, worst2, name2, tol_g2, e_all, worst, flips))
    assert e_all2 <= tol_all2, e_all2
    assert worst2 <= tol_g2, (name2, worst2)


def test_config2_window_vs_oracle(gemm_mode):
    """BASELINE config 2's window: EDVR-M nf64, 5 x 180 x 320, offsets rescaled to a mean of 1 px (the bench's default)."""
    _window_vs_oracle(64, 5, 180, 320, gemm_mode, 1.0)


def test_config3_window_vs_oracle():
    """BASELINE config 3/4's window: nf128, 7 x 180 x 320 (bf16x3 mode only: the exact-f32 mode is covered at nf64 above and at
    32 x 48 by the fixture of the reference's own class; the oracle takes ~40 s on 32 threads here)."""
    from realvsr_amd import _lib
    old = _lib.get_gemm_mode()
    _lib.set_gemm_mode('bf16x3')
    try:
        _window_vs_oracle(128, 7, 180, 320, 'bf16x3', 1.0)
    finally:
        _lib.set_gemm_mode(old)
