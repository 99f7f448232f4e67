#!/usr/bin/env python
"""bench.py -- HR frames/sec (fwd + loss + bwd + optimizer step) of the EDVR hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1: one rank per GPU over RCCL; either the driver
launches the ranks (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*) or, when WORLD_SIZE is unset,
this script re-executes itself under torch.distributed.run with N ranks (the analogue of codes/train.py:19-26).
W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() on both sides; max over
ranks; rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] ("EDVR-M 64ch, 5-frame 180x320 LR, batch 8, fwd+bwd on 1xMI355X"):
EDVR(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, w_TSA=True), x4 output, driven through the package's
train-step harness realvsr_amd.VideoSR_model.VideoSRModel.optimize_parameters -- the reference's step sequence
(VideoSR_AllPair_model_YCbCr_Split.py:163-191): zero_grad, netG, 1.0 * LapPyrLoss(3,'ssim','cb') on Y +
1.0 * GWLoss(w=4) on CbCr (the criteria 'lappyr' / 'gw' of the shipped option files), backward, Adam(0.9, 0.99).
`--lf-mode cb` swaps the (third-party, parity-unpinned) SSIM low-frequency term for the in-tree Charbonnier one.
Synthetic data: x ~ U[0,1) seed 1234, GT ~ U[0,1) seed 1235, default module init under seed 0 with
conv_offset_mask.weight ~ N(0, 0.01^2).  With that init alone the deformable offsets are ~0.004 px, which flatters the
gather (SURVEY.md 8d asks for O(1) px), so by DEFAULT the offset rows of every conv_offset_mask are rescaled until the mean
|offset| of each DCN is 1 px (`--offset-px P` for another value, `--offset-px 0` keeps the raw init; i.i.d. per pixel, i.e.
harsher than a trained, spatially smooth field); the line reports the measured offset statistics, and `offset_sweep`
carries the raw-init (~0 px) and 3 px steps.  Weak scaling: every rank processes its own B windows; value = N*B*K / time.
`--config {2,3,5}` selects a BASELINE.json configuration: 2 (default) = EDVR-M nf64 / 5 frames / B 8; 3 = nf128 / 7 frames /
B 16 (per GPU: with --gpus 8 this is BASELINE config 4); 5 = 540x960 sliding-window inference (fwd only, replicas).

Extra objects on the line:
  roofline     -- the fused DCN forward kernel (the kernel north_star grades): algorithmic bytes 4*(C+216+Co) per
                  output pixel summed over the timed DCN launches / their HIP-event durations, vs the
                  8 TB/s HBM3E peak.  `traffic`: HBM bytes per launch from PMC counters -- measured in this run when rocprofv3 is on
                  the box (two short child processes after the timed region: --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of
                  tools/dcn_micro.py, rescaled to this line's launch size), else the record kept under profiles/ (`traffic_source` says which).
  roofline_conv -- the 3x3 nf->nf convolution (conv_fwd5_kernel) on all B*N frames, timed live after the timed region: MFMA
                  work issued (3 bf16 passes per product) vs the 2.5 PFLOP/s dense bf16 peak, and the f32-equivalent rate;
                  `sustained_peak` / `frac_of_sustained`: the MFMA rate this box sustains on a register-resident MFMA stream whose
                  operands carry data (measured live; ~0.67 of nominal: the package power limit) and the kernel's fraction of THAT.
  cpu_baseline -- the CPU oracle (oracle/edvr_oracle.py, kind "port") on ONE window of the same workload (B=1),
                  1 warm-up + best-of-3, timed on this box's host cores (rank 0, N=1 only).
  parity       -- the same seeded window, same weights, through the HIP model: output / loss / every parameter gradient against
                  the oracle run that produced cpu_baseline (the oracle is the checker here, never the thing measured).
  extra        -- driver-timed side lines (rank 0, N=1): config3 (nf128 / 7 frames / B 16 training step) and config5 (10-frame
                  540x960 clip through infer.SlidingWindowRunner, eager and hipGraph), each with its own DCN-forward roofline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', str(min(os.cpu_count() or 1, 32)))  # CPU oracle threads (cpu_baseline)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0   # same guide: dense bf16 MFMA peak (no sparsity)
MFMA_F32_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32
PMC_PROFILE = {64: 'profiles/r06_dcn_fwd_pmc.json', 128: 'profiles/r06_dcn_fwd_pmc_nf128.json'}


def model_opt(args, world):
    force = bool(getattr(args, 'force_allreduce', False))
    net = dict(which_model_G='EDVR', nf=args.nf, nc=3, nframes=args.nframes, groups=8, front_RBs=5,
               back_RBs=args.back_rbs, center=None, predeblur=False, HR_in=False, w_TSA=True)
    return {'model': 'VideoSR_AllPair_YCbCr_Split', 'dist': world > 1 or force, 'gpu_ids': [0], 'is_train': True, 'scale': 4,
            'augment': None, 'network_G': net, 'path': {'pretrain_model_G': None, 'strict_load': True},
            'train': {'pixel_criterion_y': 'lappyr', 'pixel_weight_y': 1.0, 'pixel_criterion_c': 'gw',
                      'pixel_weight_c': 1.0, 'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-4, 'beta1': 0.9,
                      'beta2': 0.99, 'force_allreduce': force}}


def init_weights(net, offset_init_std=0.01):
    """Default module init under seed 0 was done by the caller; give conv_offset_mask non-zero weights (zero-init would
    make every DCN a plain conv)."""
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'conv_offset_mask.weight' in name:
                p.copy_((torch.randn(p.shape, generator=gen) * offset_init_std).to(p.device))


def dcn_packs(net):
    from realvsr_amd.archs.dcn import ModulatedDeformConvPack
    return [(n, m) for n, m in net.named_modules() if isinstance(m, ModulatedDeformConvPack)]


def offset_stats(net, x, target_px=None):
    """One no-grad forward with a pre-hook on every DCN pack: mean / max |offset| of its conv_offset_mask output.
    With target_px the OFFSET rows (first 2/3 of the output channels; the mask rows are left alone) of each pack's
    conv_offset_mask are rescaled, in forward order, so that its mean |offset| becomes target_px."""
    from realvsr_amd import functional as RF
    stats, hooks = {}, []

    def make(name, pack):
        def pre(_mod, inputs):
            feat = inputs[0][1] if pack.extra_offset_mask else inputs[0]
            n_off = pack.conv_offset_mask.out_channels * 2 // 3
            with torch.no_grad():
                off = RF.conv2d(feat, pack.conv_offset_mask)[:, :n_off]
                mean = off.abs().mean().item()
                if target_px is not None and mean > 0:
                    s = target_px / mean
                    pack.conv_offset_mask.weight[:n_off] *= s
                    pack.conv_offset_mask.bias[:n_off] *= s
                    off = off * s
                stats[name] = (off.abs().mean().item(), off.abs().max().item())
        return pre

    for name, pack in dcn_packs(net):
        hooks.append(pack.register_forward_pre_hook(make(name, pack)))
    with torch.no_grad():
        net(x)
    for h in hooks:
        h.remove()
    return stats


def make_batch(B, N, H, W, device, rank=0):
    x = torch.rand(B, N, 3, H, W, generator=torch.Generator().manual_seed(1234 + 1000 * rank))
    gt = torch.rand(B, 3, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235 + 1000 * rank))
    return x.to(device), gt.to(device)


class DcnTimer:
    """HIP events around every fused-DCN forward (and backward) launch on the launching (current) stream."""

    def __init__(self):
        self.events, self.bytes, self.bwd_events, self.lds_bytes, self.flops = [], 0.0, [], 0.0, 0.0

    def install(self):
        from realvsr_amd import functional as RF
        real = RF._lib.lib()
        orig_f, orig_b = real.rvsr_dcn_pack_forward, real.rvsr_dcn_pack_backward
        timer = self

        def timed(*a):
            # a: input, weight, bias, om, output, B, C, H, W, Co, stride, pad, dil, dg, act, slope, ws, ws_bytes, stream
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = orig_f(*a)
            e.record()
            B, C, H, W, Co, stride, pad, dil = a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]
            Ho = (H + 2 * pad - (2 * dil + 1)) // stride + 1
            Wo = (W + 2 * pad - (2 * dil + 1)) // stride + 1
            timer.events.append((s, e))
            timer.bytes += 4.0 * (C * H * W + (216 + Co) * Ho * Wo) * B + 4.0 * Co * C * 9
            timer.flops += 2.0 * C * 9 * Co * Ho * Wo * B     # the GEMM of the fused DCN, f32-equivalent
            # LDS bytes the kernel's formulation reads per output pixel: 9 taps x 4 corners x C channels x 4 B of x tile (the gather) +
            # the weight fragments a wave re-reads per (tap, 16-channel chunk): 4 x 1 KB (hi, lo of two 32-row blocks; x2 for Co = 128) per 32 pixels
            timer.lds_bytes += (9 * 4 * C * 4 + 9 * (C // 16) * 4096 * max(1, Co // 64) / 32.0) * Ho * Wo * B
            return rc

        def timed_bwd(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = orig_b(*a)
            e.record()
            timer.bwd_events.append((s, e))
            return rc

        _Proxy.wrap(RF, {'rvsr_dcn_pack_forward': timed, 'rvsr_dcn_pack_backward': timed_bwd})

    def uninstall(self):
        from realvsr_amd import functional as RF
        _Proxy.unwrap(RF)

    def result(self):
        ms = sum(s.elapsed_time(e) for s, e in self.events)
        return len(self.events), ms, self.bytes

    def backward_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.bwd_events)

    def matrix_frac(self, gemm_mode):
        """The same launches against the matrix-core roof: MFMA work issued (GEMM_PASSES bf16 products per f32 product) / time / dense peak."""
        ms = sum(s.elapsed_time(e) for s, e in self.events)
        if ms <= 0:
            return None
        peak = MFMA_F32_PEAK_TFLOPS if gemm_mode == 'f32' else MFMA_BF16_PEAK_TFLOPS
        return GEMM_PASSES[gemm_mode] * self.flops / (ms * 1e-3) / 1e12 / peak


def dcn_roofline(kernel, C, Co, gemm_mode, timer):
    """The `roofline` object of the fused DCN forward from a DcnTimer: against HBM where the issued matrix work per byte is below the ridge
    (nf64), against the matrix cores where it is above (nf128 in the 3-term format); the other fraction rides beside it."""
    nl, kms, kbytes = timer.result()
    bound, ai = dcn_bound(C, Co, gemm_mode)
    hbm = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else None
    mfrac = timer.matrix_frac(gemm_mode)
    peak_tf = MFMA_F32_PEAK_TFLOPS if gemm_mode == 'f32' else MFMA_BF16_PEAK_TFLOPS
    r = {'kernel': kernel, 'bound': bound}
    if bound == 'mfma':
        r.update({'achieved': None if mfrac is None else round(mfrac * peak_tf, 1), 'peak': peak_tf, 'unit': 'TFLOP/s',
                  'frac': None if mfrac is None else round(mfrac, 4),
                  'hbm_frac': None if hbm is None else round(hbm / HBM_PEAK_GBS, 4), 'hbm_achieved_GBs': None if hbm is None else round(hbm, 2)})
    else:
        r.update({'achieved': None if hbm is None else round(hbm, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                  'frac': None if hbm is None else round(hbm / HBM_PEAK_GBS, 4),
                  'matrix_frac': None if mfrac is None else round(mfrac, 4)})
    r.update({'issued_flop_per_byte': round(ai, 1), 'ridge_flop_per_byte': round(peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9), 1),
              'launches': nl, 'avg_launch_ms': round(kms / max(nl, 1), 4), 'algorithmic_bytes_per_launch': round(kbytes / max(nl, 1))})
    return r


def dcn_bound(C, Co, gemm_mode):
    """Which roof the fused DCN forward sits under: arithmetic intensity of the issued matrix work, passes * 2 C 9 Co FLOP over
    4 (C + 216 + Co) B per output pixel, against the ridge peak_flops / 8 TB/s.  nf64 in the 3-term format: 161 < 312 -> HBM; nf128: 469 -> MFMA."""
    peak = MFMA_F32_PEAK_TFLOPS if gemm_mode == 'f32' else MFMA_BF16_PEAK_TFLOPS
    ai = GEMM_PASSES[gemm_mode] * 2.0 * C * 9 * Co / (4.0 * (C + 216 + Co))
    return ('mfma' if ai > peak * 1e12 / (HBM_PEAK_GBS * 1e9) else 'hbm'), ai


class _Proxy:
    """Minimal proxy so C-ABI symbols can be intercepted without touching the library object."""

    def __init__(self, lib, fns):
        self.__dict__['_lib'], self.__dict__['_fns'] = lib, fns

    def __getattr__(self, k):
        fn = self._fns.get(k)
        return fn if fn is not None else getattr(self._lib, k)

    @staticmethod
    def wrap(RF, fns):
        real = RF._lib.lib()
        RF._lib._lib = _Proxy(real, fns)
        _Proxy._real = real

    @staticmethod
    def unwrap(RF):
        RF._lib._lib = _Proxy._real


def sustained_mfma_tflops(dev, pattern='split'):
    """The bf16 MFMA rate this box SUSTAINS on operands that carry data (measured live, ~0.1 s): a register-resident stream of
    v_mfma_f32_32x32x16_bf16 (rvsr_debug_mfma_rate: nothing but the matrix pipe runs) on `pattern` operands: 'ones' (constant: the
    nominal peak), 'normal' (random sign / exponent / mantissa) or 'split' (the bf16x3 kernels' operand mix: N(0,1)-like hi parts and
    lo parts ~2^-9 of them).  With data the package power limit, not the kernel, caps the matrix pipe at ~0.67 of 2.5 PFLOP/s
    (profiles/r05_mfma_power_micro.txt): the second denominator reported next to the guide's nominal peak."""
    from realvsr_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(1)
    n = 8 * 512 * 8
    if pattern == 'ones':
        ops = torch.ones(n)
    else:
        ops = torch.randn(n, generator=g)
        if pattern == 'split':
            ops.view(8, -1)[1::2] *= 2.0 ** -9
    ops = ops.to(torch.bfloat16).to(dev)
    out = torch.empty(256 * 512, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    iters = 150000
    L.rvsr_debug_mfma_rate(ops.data_ptr(), out.data_ptr(), 256, 2000, st)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = L.rvsr_debug_mfma_rate(ops.data_ptr(), out.data_ptr(), 256, iters, st)
    e.record()
    torch.cuda.synchronize()
    if rc != 0:
        return None
    return 256 * 8 * iters * 8 * 32768.0 / (s.elapsed_time(e) * 1e-3) / 1e12


def measure_dcn_traffic(nf, timeout=150):
    """HBM bytes per output pixel of the fused DCN forward, MEASURED in this run: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate
    runs, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) of tools/dcn_micro.py at the L1 shape (B*N frames of
    180x320, offset std 1.25 px), each in its own process.  gfx950: FETCH_SIZE x 2 (it reports half of the coalesced reads, profiles/r01_notes.md),
    both counters in KB.  Returns (bytes_per_pixel, note) or (None, reason); the caller falls back to the record kept under profiles/."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    B = 40 if nf == 64 else 16
    vals = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='rvsr_pmc_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            subprocess.run([exe, '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '--', sys.executable,
                            os.path.join(ROOT, 'tools', 'dcn_micro.py'), '--iters', '2', '--B', str(B), '--C', str(nf), '--ostd', '1.25', '--fwd-only'],
                           cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=False)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            per = {}
            for fn in files:
                with open(fn) as f:
                    for row in csv.DictReader(f):
                        name = row.get('Kernel_Name', '')
                        if 'dcn_fwd3' in name and row.get('Counter_Name') == ctr:
                            acc = per.setdefault(name, [0.0, 0])
                            acc[0] += float(row['Counter_Value'])
                            acc[1] += 1
            if not per:
                return None, 'no %s rows for dcn_fwd3 in the rocprofv3 output' % ctr
            vals[ctr] = max(v / n for v, n in per.values())   # (the halo candidates that return at once count ~0: keep the one that worked)
        except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
            return None, '%s pass failed: %s' % (ctr, type(e).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    px = B * 180 * 320
    hbm = (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
    return hbm / px, ('measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate processes, --kernel-trace) of '
                      'tools/dcn_micro.py --B %d --C %d --ostd 1.25 --fwd-only; FETCH_SIZE x 2 + WRITE_SIZE = %.1f B per output pixel, rescaled to '
                      "this line's average launch size" % (B, nf, hbm / px))


def conv_roofline(net, frames, nf, H, W, gemm_mode, reps=10, sustained=True):
    """Second roofline object (outside the timed region): the 3x3 nf->nf convolution of the feature extractor on all B*N
    frames -- the shape ~80 % of a step's GEMM work runs at -- timed live with HIP events on the launching stream.  `achieved`
    counts the MFMA work actually issued (3 bf16 passes per f32 product in bf16x3 mode) against the dense bf16 peak;
    `f32_equivalent` is the algorithmic 2*M*N*K rate."""
    from realvsr_amd import functional as RF
    conv = net.feature_extraction[0].conv1
    x = torch.randn(frames, nf, H, W, device=conv.weight.device)
    with torch.no_grad():
        for _ in range(10):
            RF.conv2d(x, conv, act=RF.ACT_RELU)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            RF.conv2d(x, conv, act=RF.ACT_RELU)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    flop = 2.0 * frames * H * W * nf * nf * 9
    passes = GEMM_PASSES[gemm_mode]
    peak = MFMA_F32_PEAK_TFLOPS if gemm_mode == 'f32' else MFMA_BF16_PEAK_TFLOPS
    ach = passes * flop / (ms * 1e-3) / 1e12
    r = {'kernel': 'conv_fwd5_kernel (+ its weight pre-pack): 3x3 %d->%d + ReLU on %d frames of %dx%d' % (nf, nf, frames, H, W),
         'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
         'mfma_passes_per_product': passes, 'f32_equivalent': round(flop / (ms * 1e-3) / 1e12, 1),
         'avg_launch_ms': round(ms, 4), 'traffic': None}
    if gemm_mode != 'f32' and sustained:   # (skipped with --no-sweep: the rocprofv3 kernel statistics then hold the step's kernels only)
        sus = sustained_mfma_tflops(conv.weight.device, 'split' if gemm_mode != 'bf16' else 'normal')
        ones = sustained_mfma_tflops(conv.weight.device, 'ones')
        if sus:
            r['sustained_peak'] = round(sus, 1)
            r['frac_of_sustained'] = round(ach / sus, 4)
            r['sustained_peak_note'] = ('measured live on this box: register-resident v_mfma_f32_32x32x16_bf16 stream on operands with this '
                                        "mode's data pattern (package power limit); the same stream on constant operands: %s TFLOP/s"
                                        % (round(ones, 1) if ones else None))
    return r


class _Snapshot:
    """Parameters + Adam state of the model, to put back after an excursion (offset rescale, other GEMM mode)."""

    def __init__(self, model):
        o = model.optimizer_G
        self.o = o
        self.param, self.m1, self.m2 = o.buffers.param.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone()
        self.steps = list(o._steps)

    def restore(self):
        o = self.o
        o.buffers.param.copy_(self.param)
        o.exp_avg.copy_(self.m1)
        o.exp_avg_sq.copy_(self.m2)
        o._steps = list(self.steps)
        for t, v in zip(o._step_t, self.steps):
            t.fill_(float(v))
        from realvsr_amd import functional as RF
        RF.packed_weights.repack()   # the flat parameter buffer was overwritten behind torch's version counters


def timed_steps(model, n, first_step):
    """3 untimed + n timed optimizer steps with the DCN timers installed: (ms/step, DCN fwd frac of the HBM peak, DCN bwd ms/step).
    (Three untimed steps: the first one after a change of offsets / GEMM mode re-sizes workspaces and allocator pools, and the DCN forward
    picks its tile halo from the offset counters of three optimizer steps ago, functional.DcnOffsetStats.LAG.)"""
    for _ in range(3):
        model.optimize_parameters(first_step, log=False)
    timer = DcnTimer()
    timer.install()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        model.optimize_parameters(first_step + 1 + i, log=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.uninstall()
    nl, kms, kbytes = timer.result()
    frac = kbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if kms > 0 else None
    from realvsr_amd import _lib as rlib
    timed_steps.last_matrix_frac = timer.matrix_frac(rlib.get_gemm_mode())
    return 1e3 * dt / n, frac, timer.backward_ms() / n


def offset_sweep(model, x, first_step, native, pxs=(3.0,), steps=3):
    """SURVEY.md 8d "second large-motion set", on the driver-timed line: the same step with the raw conv_offset_mask init (`native`:
    the rows saved before the default rescale, mean |offset| ~0.004 px -- the flattering end) and with every conv_offset_mask rescaled to
    a mean |offset| of P px (i.i.d. per pixel: harsher than trained, spatially smooth fields), 3 untimed + `steps` timed steps each,
    parameters and optimizer state restored afterwards.  dcn_bwd_ms = HIP-event time of the fused DCN backward calls (input/offset/mask
    gradient + weight gradient kernels) per step."""
    from realvsr_amd import functional as RF
    out = {}
    snap = _Snapshot(model)

    def entry(key):
        ms, frac, bwd = timed_steps(model, steps, first_step)
        st = offset_stats(model.netG, x)
        l1 = st.get('pcd_align.L1_dcnpack', (None, None))
        out[key] = {'ms_per_step': round(ms, 3), 'dcn_fwd_frac': None if frac is None else round(frac, 4),
                    'dcn_bwd_ms': round(bwd, 3), 'steps': steps,
                    'offset_abs_mean_px': None if l1[0] is None else round(l1[0], 3)}
        snap.restore()

    if native:
        with torch.no_grad():
            for name, pack in dcn_packs(model.netG):
                w, b = native[name]
                pack.conv_offset_mask.weight.copy_(w)
                pack.conv_offset_mask.bias.copy_(b)
        RF.packed_weights.repack()
        entry('raw_init')
    for P in pxs:
        offset_stats(model.netG, x, P)
        entry('%gpx' % P)
    return out


def gemm_mode_step(model, first_step, mode, steps=2):
    """The same step in another GEMM mode ('f32': v_mfma_f32_32x32x2_f32 everywhere; 'bf16x2' / 'bf16': the reduced-term speed modes),
    2 untimed + `steps` timed; parameters and optimizer state are put back afterwards."""
    from realvsr_amd import _lib as rlib
    from realvsr_amd import functional as RF
    snap = _Snapshot(model)
    old = rlib.get_gemm_mode()
    rlib.set_gemm_mode(mode)
    try:
        ms, frac, bwd = timed_steps(model, steps, first_step)
    finally:
        rlib.set_gemm_mode(old)
        snap.restore()
        RF.invalidate_weight_cache()
    return {'ms_per_step': round(ms, 3), 'dcn_fwd_frac': None if frac is None else round(frac, 4), 'dcn_bwd_ms_per_step': round(bwd, 3)}


def f32_mode_step(model, first_step, steps=2):
    return gemm_mode_step(model, first_step, 'f32', steps)['ms_per_step']


GEMM_DESC = {'bf16x3': ' (3-term bf16 split on v_mfma_f32_32x32x16_bf16, f32 accumulate)',
             'bf16x2': ' (2-term bf16 split: weights rounded to bf16, activations hi + lo; v_mfma_f32_32x32x16_bf16, f32 accumulate)',
             'bf16': ' (bf16 operands on v_mfma_f32_32x32x16_bf16, f32 accumulate)',
             'f16fp8': ' (forward 3x3 convs: f16 main term + fp8 e4m3 cross terms on v_mfma_f32_32x32x16_f16 / v_mfma_scale_f32_32x32x64_f8f6f4, '
                       '~1.2e-5 per conv instead of 4.6e-6; gradients and every other kernel: the 3-term bf16 split)',
             'f32': ' (v_mfma_f32_32x32x2_f32, exact f32)'}
GEMM_PASSES = {'bf16x3': 3, 'bf16x2': 2, 'bf16': 1, 'f32': 1, 'f16fp8': 3}


def dtype_of(gemm_mode):
    """Tensors, accumulation and all non-GEMM arithmetic are f32 in every mode; the product format is the GEMM mode."""
    return 'f32' if gemm_mode == 'f32' else 'f32 (%s GEMM)' % gemm_mode


def cpu_baseline(args, sd_in=None):
    """One window (B=1) of the same workload through the CPU oracle: fwd + loss + bwd; BASELINE.md section 2 protocol
    (1 warm-up, then best of 3).  `sd_in`: the benchmarked model's state_dict (so the CPU side runs the weights -- and the
    offsets -- the GPU side was timed on); default: the module init.  Returns (json object, oracle results of the last run) --
    the second feeds `parity_check`."""
    from oracle import edvr_oracle as O
    from realvsr_amd.archs.EDVR_arch import EDVR
    nf, N, H, W = args.nf, args.nframes, args.height, args.width
    if sd_in is None:
        torch.manual_seed(0)
        net = EDVR(nf=nf, nc=3, nframes=N, groups=8, front_RBs=5, back_RBs=args.back_rbs, w_TSA=True)
        init_weights(net)
        sd_in = net.state_dict()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sd_in.items()}
    x = torch.rand(1, N, 3, H, W, generator=torch.Generator().manual_seed(1234))
    gt = torch.rand(1, 3, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235))
    last = {}

    def one():
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.edvr_forward(sd, x, nframes=N, groups=8, front_RBs=5, back_RBs=args.back_rbs, w_TSA=True)
        loss = O.lap_pyr_loss(out[:, 0:1], gt[:, 0:1], 3, lf_mode=args.lf_mode) + O.gw_loss(out[:, 1:3], gt[:, 1:3], 4)
        loss.backward()
        dt = time.perf_counter() - t0
        last['out'], last['loss'] = out.detach(), float(loss.item())
        return dt

    # Thread count: min(hardware threads, 32).  The torch CPU conv path collapses from oversubscription on this host: the same
    # window took 432 s with all 256 hardware threads and 8.4 s with 32 (measured by an earlier version of this function, see
    # profiles/r02_notes.md), so "all cores" would both misrepresent the CPU and blow the bench's time budget.
    ncpu = os.cpu_count() or 1
    cores = min(ncpu, 32)
    torch.set_num_threads(cores)
    one()                                       # warm-up
    best = min(one() for _ in range(3))
    last['grads'] = {k: v.grad.detach() for k, v in sd.items() if v.grad is not None}
    last['x'], last['gt'] = x, gt
    cpu_model = ''
    try:
        with open('/proc/cpuinfo') as f:
            cpu_model = next((l.split(':', 1)[1].strip() for l in f if l.startswith('model name')), '')
    except OSError:
        pass
    return {'value': round(1.0 / best, 5), 'unit': 'HR frames/s', 'cores': cores, 'kind': 'port',
            'sample': '1 window (B=1, %d frames %dx%d LR) fwd+loss+bwd through oracle/edvr_oracle.py (torch %s CPU ops + '
                      'OpenMP C DCN) on the weights the GPU step was timed on; 1 warm-up + best of 3 = %.2f s; %d threads on a '
                      'host with %d hardware threads (%s)' % (N, H, W, torch.__version__, best, cores, ncpu, cpu_model)}, last


def parity_check(model, ora):
    """The oracle's window (same seeded input, same weights) through the HIP model: forward, the model's own criteria, backward
    (no optimizer step); output / loss / per-parameter gradients against the oracle's.  Errors are relative L2 norms
    (|a - b| / |b|); `out_max_abs_err` is the largest element-wise difference of the [0, 1]-range output."""
    dev = model.device
    keep_L, keep_H = getattr(model, 'var_L', None), getattr(model, 'var_H', None)
    model.feed_data({'LQs': ora['x'].to(dev), 'GT': ora['gt'].to(dev)})
    if model.reducer is not None:
        model.reducer.zero_grad()
    else:
        model.optimizer_G.zero_grad()
    total, _ = model.forward_loss()
    total.backward()
    if model.reducer is not None:
        model.reducer.finish()
    torch.cuda.synchronize()

    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / (b.norm() + 1e-300))

    out = model.fake_H
    worst, worst_name, num, den, missing = 0.0, None, 0.0, 0.0, []
    for k, p in model.netG.named_parameters():
        g = ora['grads'].get(k)
        if g is None or p.grad is None:
            missing.append(k)
            continue
        e = l2(p.grad, g)
        num += float((p.grad.detach().double().cpu() - g.double()).pow(2).sum())
        den += float(g.double().pow(2).sum())
        if e > worst:
            worst, worst_name = e, k
    # Two scale-free views of the output error.  (1) against the RESIDUAL branch: the output is bilinear(centre frame) + residual and a
    # freshly initialised network's residual is small, so out_rel_err flatters; (2) what the error would do to a 30 dB model: PSNR-Y of
    # both outputs against a synthetic target = oracle output + N(0, 10^(-30/20)) -- the north star's bound is 1e-3 dB
    import torch.nn.functional as F
    oc, oo = out.detach().float().cpu(), ora['out'].float()
    base = F.interpolate(ora['x'][:, ora['x'].shape[1] // 2].float(), scale_factor=oo.shape[-1] // ora['x'].shape[-1], mode='bilinear', align_corners=False)
    noise = torch.randn(oo[:, 0].shape, generator=torch.Generator().manual_seed(4321)) * 10 ** (-30 / 20)
    target = oo[:, 0] + noise

    def psnr(a):
        return float(10 * torch.log10(1.0 / (a.double() - target.double()).pow(2).mean()))
    res = {'window': 'B=1, seeds 1234/1235, weights of the timed model', 'out_rel_err': float('%.3e' % l2(out, ora['out'])),
           'residual_rel_err': float('%.3e' % float((oc - oo).double().norm() / ((oo - base).double().norm() + 1e-300))),
           'psnr_y_delta_db_at_30db': float('%.3e' % abs(psnr(oc[:, 0]) - psnr(oo[:, 0]))),
           'out_max_abs_err': float('%.3e' % (out.detach().cpu() - ora['out']).abs().max().item()),
           'loss_rel_err': float('%.3e' % (abs(float(total.item()) - ora['loss']) / max(abs(ora['loss']), 1e-30))),
           'grad_l2_err_all': float('%.3e' % ((num / max(den, 1e-300)) ** 0.5)), 'worst_param': worst_name,
           'worst_param_l2_err': float('%.3e' % worst), 'params_compared': len(ora['grads']) - len(missing),
           'params_without_gradient': missing, 'checker': 'oracle/edvr_oracle.py (CPU)'}
    model.optimizer_G.zero_grad()
    if keep_L is not None:
        model.var_L, model.var_H = keep_L, keep_H
    return res


class _Cfg:
    """argparse-like bundle for the side lines."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def extra_train_line(base, nf, nframes, batch, steps=3):
    """A second training configuration timed on the driver's box after the main line (BASELINE config 3: nf128 / 7 frames / B 16):
    same harness, same offset protocol, 3 untimed + `steps` timed steps."""
    from realvsr_amd import loss as L
    from realvsr_amd.VideoSR_model import create_model
    a = _Cfg(**dict(vars(base), nf=nf, nframes=nframes, batch=batch, force_allreduce=False))
    dev = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(0)
    model = create_model(model_opt(a, 1))
    init_weights(model.netG)
    if a.lf_mode == 'cb':
        model.cri_pix_y = L.LapPyrLoss(3, 'cb', 'cb', 'mean')
    x, gt = make_batch(batch, nframes, a.height, a.width, dev)
    if a.offset_px:
        offset_stats(model.netG, x, a.offset_px)
    model.feed_data({'LQs': x, 'GT': gt})
    ms, frac, bwd = timed_steps(model, steps, 1)
    mfrac = timed_steps.last_matrix_frac
    st = offset_stats(model.netG, x)
    l1 = st.get('pcd_align.L1_dcnpack', (None, None))
    from realvsr_amd import _lib as rlib
    conv = conv_roofline(model.netG, batch * nframes, nf, a.height, a.width, rlib.get_gemm_mode(), reps=3)
    out = {'workload': 'EDVR nf%d, %d-frame %dx%d LR windows, batch %d, fwd + loss + bwd + Adam step' % (nf, nframes, a.height, a.width, batch),
           'ms_per_step': round(ms, 2), 'value': round(batch / (ms * 1e-3), 3), 'unit': 'HR frames/s', 'steps': steps,
           'dcn_fwd_frac': None if frac is None else round(frac, 4), 'dcn_bwd_ms_per_step': round(bwd, 2),
           'dcn_fwd_bound': dcn_bound(nf, nf, rlib.get_gemm_mode())[0],
           'dcn_fwd_matrix_frac': None if mfrac is None else round(mfrac, 4),
           'offset_abs_mean_px': None if l1[0] is None else round(l1[0], 3), 'conv_frac_of_bf16_peak': conv['frac'],
           'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if rlib.get_gemm_mode() == 'bf16x3':   # the opt-in speed mode on the same workload (2 untimed + 2 timed steps)
        r1 = gemm_mode_step(model, steps + 4, 'bf16', steps=2)
        out['speed_mode_bf16'] = dict(r1, value=round(batch / (r1['ms_per_step'] * 1e-3), 3))
    del model, x, gt
    return out


def infer_line(a, T=10, offset_px=None):
    """BASELINE config 5: a T-frame 540x960 clip through infer.SlidingWindowRunner (per-frame feature reuse), forward only, eager and as
    a hipGraph replay of the window stage; ms per 2160x3840 output frame and the DCN-forward roofline of the eager pass."""
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.infer import SlidingWindowRunner
    dev = torch.device('cuda', torch.cuda.current_device())
    torch.manual_seed(0)
    net = EDVR(nf=a.nf, nc=3, nframes=a.nframes, groups=8, front_RBs=5, back_RBs=a.back_rbs, w_TSA=True)
    init_weights(net)
    net = net.to(dev).eval()
    clip = torch.rand(T, 3, a.height, a.width, generator=torch.Generator().manual_seed(1234)).to(dev)
    if offset_px:
        offset_stats(net, clip[:a.nframes].unsqueeze(0).contiguous(), offset_px)
    st = offset_stats(net, clip[:a.nframes].unsqueeze(0).contiguous())
    l1 = st.get('pcd_align.L1_dcnpack', (None, None))

    def timed(run):
        run(clip)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(clip)
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) * 1e3 / T

    run = SlidingWindowRunner(net, a.nframes, padding='replicate', chunk=2)
    out, ms_eager = timed(run)
    timer = DcnTimer()
    timer.install()
    run(clip)
    torch.cuda.synchronize()
    timer.uninstall()
    nl, kms, kbytes = timer.result()
    from realvsr_amd import _lib as rlib
    mfrac = timer.matrix_frac(rlib.get_gemm_mode())
    run_g = SlidingWindowRunner(net, a.nframes, padding='replicate', chunk=2, use_graph=True)
    out_g, ms_graph = timed(run_g)
    res = {'workload': 'EDVR nf%d, %d-frame windows over a %d-frame %dx%d clip -> %dx%d, forward only, per-frame feature reuse'
                       % (a.nf, a.nframes, T, a.height, a.width, 4 * a.height, 4 * a.width),
           'ms_per_frame': round(min(ms_eager, ms_graph), 2), 'ms_per_frame_eager': round(ms_eager, 2),
           'ms_per_frame_hipgraph': round(ms_graph, 2), 'value': round(1e3 / min(ms_eager, ms_graph), 3), 'unit': 'HR frames/s',
           'graph_bit_identical': bool(torch.equal(out, out_g)),
           'dcn_fwd_frac': round(kbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kms > 0 else None,
           'dcn_fwd_bound': dcn_bound(a.nf, a.nf, rlib.get_gemm_mode())[0],
           'dcn_fwd_matrix_frac': None if mfrac is None else round(mfrac, 4),
           'dcn_fwd_avg_launch_ms': round(kms / max(nl, 1), 4), 'offset_abs_mean_px': None if l1[0] is None else round(l1[0], 3),
           'note': 'hipGraph path: a ring of N feature slots (one slot overwritten per frame) + one captured graph per rotation of the ring, '
                   'replayed round-robin; the eager pass is GPU-bound at this frame size (no launch gaps), so the replay has nothing to recover',
           'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    if rlib.get_gemm_mode() == 'bf16x3':   # the opt-in speed mode on the same clip (eager)
        rlib.set_gemm_mode('bf16')
        try:
            out1, ms1 = timed(run)
        finally:
            rlib.set_gemm_mode('bf16x3')
        res['speed_mode_bf16'] = {'ms_per_frame': round(ms1, 2), 'value': round(1e3 / ms1, 3),
                                  'max_abs_diff_of_the_output': float('%.3e' % (out1 - out).abs().max().item())}
        del out1
        rlib.set_gemm_mode('f16fp8')   # forward 3x3 convs in the f16 + fp8 product format, everything else three-term
        try:
            out1, ms1 = timed(run)
        finally:
            rlib.set_gemm_mode('bf16x3')
        res['speed_mode_f16fp8'] = {'ms_per_frame': round(ms1, 2), 'value': round(1e3 / ms1, 3),
                                    'max_abs_diff_of_the_output': float('%.3e' % (out1 - out).abs().max().item())}
        del out1
    del net, clip, out, out_g, run, run_g
    return res


def dry_run_plan(args):
    """`--dry-run` without a launcher: everything the N-rank run is going to do that can be known on the host -- the per-rank shard of the
    global batch (the reference: batch_size // world_size, codes/data/__init__.py:10-15; DistIterSampler's contiguous per-rank ranges,
    codes/data/data_sampler.py:46-59), the bucket table of the gradient all-reduce built by the SAME code the run uses
    (realvsr_amd.dist.BucketedGradAllReduce on the architecture's parameters), the ring time of each bucket on one xGMI link, and the memory
    plan (4 flat buffers + the last measured single-GPU peak of this configuration).  Needs no GPU."""
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.dist import BucketedGradAllReduce, shard_range
    from realvsr_amd.optim import FlatBuffers
    world, B = args.gpus, args.batch
    torch.manual_seed(0)
    net = EDVR(nf=args.nf, nc=3, nframes=args.nframes, groups=8, front_RBs=5, back_RBs=args.back_rbs, w_TSA=True)
    names = {p: n for n, p in net.named_parameters()}
    buffers = FlatBuffers([list(net.parameters())])
    bucket_mb = float(os.environ.get('RVSR_BUCKET_MB', '4'))
    red = BucketedGradAllReduce(None, bucket_mb=bucket_mb, buffers=buffers, broadcast=False)
    link_gbs = 153.0   # one xGMI link direction (MI355X_MICROARCH / SURVEY.md section 5): a ring all-reduce is per-link bound
    table = []
    for i, (s, e) in enumerate(red.buckets):
        members = [q for q in red.params if red._bucket_of[q] == i]
        table.append({'bucket': i, 'offset_elems': s, 'bytes': 4 * (e - s), 'params': len(members), 'first': names[members[0]],
                      'last': names[members[-1]],
                      'ring_ms_on_one_link': round(2.0 * (world - 1) / max(world, 1) * 4 * (e - s) / (link_gbs * 1e9) * 1e3, 4)})
    measured = None
    for r in range(9, 0, -1):   # the newest committed driver-style line that carries this configuration's peak
        try:
            with open(os.path.join(ROOT, 'profiles', 'r%02d_bench_default.json' % r)) as f:
                c3 = json.load(f).get('extra', {}).get('config3', {})
            if args.nf == 128 and args.nframes == 7 and c3.get('peak_mem_GB'):
                measured = {'peak_mem_GB': c3['peak_mem_GB'], 'batch': 16, 'source': 'profiles/r%02d_bench_default.json extra.config3' % r}
                break
        except (OSError, ValueError):
            continue
    flat_gb = 4 * 4 * buffers.numel / 2 ** 30
    mem = {'flat_buffers_GB': round(flat_gb, 3), 'hbm_per_gpu_GB': 288,
           'note': 'parameters, gradients and both Adam moments are four flat f32 buffers; activations dominate and scale with the per-GPU batch'}
    if measured is not None:
        est = measured['peak_mem_GB'] * B / measured['batch']
        mem.update(measured_single_gpu=measured, estimate_GB_at_this_per_gpu_batch=round(est, 1), headroom_GB=round(288 - est, 1))
    return {'dry_run': True, 'config': args.config, 'world': world, 'per_gpu_batch': B, 'global_batch': world * B,
            'arch': 'EDVR nf%d, %d frames, back_RBs %d, TSA' % (args.nf, args.nframes, args.back_rbs),
            'shards': [{'rank': r, 'windows': list(shard_range(world * B, r, world))} for r in range(world)],
            'parameters': sum(p.numel() for p in net.parameters()), 'flat_elems_incl_alignment': buffers.numel,
            'gradient_bytes': 4 * buffers.numel, 'bucket_mb': bucket_mb, 'buckets': table,
            'collectives_per_step': len(table),
            'ring_ms_total_on_one_link': round(sum(t['ring_ms_on_one_link'] for t in table), 4),
            'memory': mem, 'backend': os.environ.get('RVSR_BENCH_BACKEND', 'nccl'),
            'launch': 'python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py '
                      '--config %d --gpus %d' % (world, args.config, world)}


def relaunch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--nf', type=int, default=64)
    ap.add_argument('--nframes', type=int, default=5)
    ap.add_argument('--back-rbs', type=int, default=10)
    ap.add_argument('--height', type=int, default=180)
    ap.add_argument('--width', type=int, default=320)
    ap.add_argument('--lf-mode', choices=['ssim', 'cb'], default='ssim')
    ap.add_argument('--offset-px', type=float, default=1.0,
                    help='rescale every conv_offset_mask so that the mean |offset| of its DCN is this many pixels (default 1; 0 = '
                         'keep the raw N(0, 0.01^2) init, ~0.004 px)')
    ap.add_argument('--config', type=int, choices=[2, 3, 5], default=2,
                    help='BASELINE.json configuration: 2 = nf64 / 5 frames / B 8 (default), 3 = nf128 / 7 frames / B 16 per GPU '
                         '(with --gpus 8: config 4), 5 = 540x960 sliding-window inference')
    ap.add_argument('--no-extra', action='store_true', help='skip the config-3 / config-5 side lines')
    ap.add_argument('--dry-run', action='store_true', help='N > 1: check launcher env, device count and backend, then exit')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sweep', action='store_true', help='skip the post-run offset sweep and the f32-mode step')
    ap.add_argument('--force-allreduce', action='store_true',
                    help='N = 1: initialise a one-rank RCCL process group and run the bucketed gradient all-reduce (hooks, one async '
                         'all_reduce per bucket, finish) inside every step -- arithmetically the identity, but it loads librccl and '
                         'exercises the overlap path on a 1-GPU box; the line then carries the `allreduce` object')
    pre, _ = ap.parse_known_args()
    if pre.config == 3:
        ap.set_defaults(nf=128, nframes=7, batch=16)
    elif pre.config == 5:
        ap.set_defaults(nf=128, nframes=7, batch=1, height=540, width=960)
    args = ap.parse_args()
    if args.offset_px is not None and args.offset_px <= 0:
        args.offset_px = None

    if args.dry_run and 'WORLD_SIZE' not in os.environ:
        print(json.dumps(dry_run_plan(args)), flush=True)   # host-only: works without a GPU and without a launcher
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    ndev = torch.cuda.device_count()
    backend = os.environ.get('RVSR_BENCH_BACKEND', 'nccl')  # 'gloo': developer check of the N>1 path on one GPU
    if backend == 'nccl' and world > ndev:
        raise SystemExit('bench.py --gpus %d: WORLD_SIZE=%d ranks over RCCL need one MI355X each, but this box shows %d GPU(s) '
                         '(HIP_VISIBLE_DEVICES=%s).  Run on a node with >= %d GPUs, or set RVSR_BENCH_BACKEND=gloo to exercise the '
                         'N > 1 code path with all ranks sharing GPU 0 (developer check, not a scaling number).'
                         % (args.gpus, world, ndev, os.environ.get('HIP_VISIBLE_DEVICES', '<unset>'), world))
    if args.dry_run:
        if rank == 0:
            print(json.dumps({'dry_run': True, 'world': world, 'visible_gpus': ndev, 'backend': backend,
                              'master': '%s:%s' % (os.environ.get('MASTER_ADDR', '127.0.0.1'), os.environ.get('MASTER_PORT', '29511')),
                              'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
                              'bucket_mb': float(os.environ.get('RVSR_BUCKET_MB', '4'))}), flush=True)
        return
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    use_pg = world > 1 or args.force_allreduce
    if use_pg:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus

    from realvsr_amd import loss as L
    from realvsr_amd import _lib as rlib
    from realvsr_amd.VideoSR_model import create_model
    gemm_mode = rlib.get_gemm_mode()
    B, N, H, W = args.batch, args.nframes, args.height, args.width
    if args.config == 5:
        # inference replicas: every rank super-resolves its own clip, no collective on the data path
        res = infer_line(args, T=10, offset_px=args.offset_px)
        t = torch.tensor([res['ms_per_frame']], device=device, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            ms = t.item()
            print(json.dumps({'metric': 'HR frames/sec (fwd only, sliding window) on %d-frame %dx%d LR windows' % (N, H, W),
                              'value': round(world * 1e3 / ms, 3), 'unit': 'HR frames/s', 'n_gpus': world, 'steps': 10, 'warmup': 10,
                              'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                              'dtype': dtype_of(gemm_mode), 'data': 'synthetic',
                              'config': {'workload': res['workload'], 'parallelism': 'replicas x%d' % world, 'gemm': gemm_mode,
                                         'offset_abs_mean_px': res['offset_abs_mean_px']},
                              'roofline': ({'kernel': 'fused DCN forward', 'bound': 'mfma', 'frac': res['dcn_fwd_matrix_frac'],
                                            'peak': MFMA_BF16_PEAK_TFLOPS if gemm_mode != 'f32' else MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                            'achieved': None if res['dcn_fwd_matrix_frac'] is None else round(
                                                res['dcn_fwd_matrix_frac'] * (MFMA_BF16_PEAK_TFLOPS if gemm_mode != 'f32' else MFMA_F32_PEAK_TFLOPS), 1),
                                            'hbm_frac': res['dcn_fwd_frac'], 'avg_launch_ms': res['dcn_fwd_avg_launch_ms'], 'traffic': None}
                                           if res['dcn_fwd_bound'] == 'mfma' else
                                           {'kernel': 'fused DCN forward', 'bound': 'hbm', 'frac': res['dcn_fwd_frac'], 'peak': HBM_PEAK_GBS,
                                            'unit': 'GB/s', 'achieved': None if res['dcn_fwd_frac'] is None else round(res['dcn_fwd_frac'] * HBM_PEAK_GBS, 1),
                                            'matrix_frac': res['dcn_fwd_matrix_frac'], 'avg_launch_ms': res['dcn_fwd_avg_launch_ms'], 'traffic': None}),
                              'detail': res}), flush=True)
        if use_pg:
            torch.distributed.destroy_process_group()
        return
    torch.manual_seed(0 if rank == 0 else 12345 + rank)   # ranks > 0 start from DIFFERENT weights on purpose:
    model = create_model(model_opt(args, world))           # the model broadcasts rank 0's (checked below)
    if rank == 0:
        init_weights(model.netG)
    if world > 1:
        from realvsr_amd.dist import broadcast_parameters
        broadcast_parameters(model.netG)
        probe = model.optimizer_G.buffers.param.double().sum()
        lo, hi = probe.clone(), probe.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert lo.item() == hi.item(), 'ranks do not hold identical parameters after the broadcast'
    if args.lf_mode == 'cb':
        model.cri_pix_y = L.LapPyrLoss(3, 'cb', 'cb', 'mean')
    x, gt = make_batch(B, N, H, W, device, rank)
    native = {name: (pack.conv_offset_mask.weight.detach().clone(), pack.conv_offset_mask.bias.detach().clone())
              for name, pack in dcn_packs(model.netG)}   # the raw init, for the sweep's ~0 px entry
    if args.offset_px is not None and rank == 0:
        offset_stats(model.netG, x, args.offset_px)
    if args.offset_px is not None and world > 1:
        broadcast_parameters(model.netG)
    model.feed_data({'LQs': x, 'GT': gt})

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        model.optimize_parameters(i + 1, log=False)
    off = offset_stats(model.netG, x) if rank == 0 else {}
    if model.reducer is not None:
        model.reducer.reset_stats()
    timer = DcnTimer()
    timer.install()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.optimize_parameters(args.warmup + i + 1, log=False)
    fence()
    dt = time.perf_counter() - t0
    timer.uninstall()
    loss = model.loss_terms['l_pix']
    allreduce = None
    if use_pg:
        # every rank must hold bit-identical parameters after the last step (same averaged gradients, same Adam arithmetic)
        probe = model.optimizer_G.buffers.param.double().sum()
        lo, hi = probe.clone(), probe.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert lo.item() == hi.item(), 'ranks diverged: parameters differ after %d steps' % (args.warmup + args.steps)
        r = model.reducer
        allreduce = {'buckets': len(r.buckets), 'bytes': 4 * r.buffers.numel, 'bucket_bytes': [4 * (e - s) for s, e in r.buckets],
                     'issued_during_backward': r.stats_issued_in_backward, 'exposed_ms': round(r.exposed_ms(), 3),
                     'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'world': world,
                     'params_identical_after_last_step': True,
                     'note': 'exposed_ms = stream time between entering finish() and the last bucket being ready, per step, '
                             'rank 0, averaged over the timed steps; issued_during_backward = buckets whose all-reduce was '
                             'enqueued from a gradient hook, i.e. before backward returned (last step)'}
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    nl, kms, kbytes = timer.result()
    # HBM traffic of the DCN forward kernel: NOT measured here.  PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3
    # runs) are kept under profiles/; their bytes per output pixel are rescaled to the pixels an average timed launch covers.
    traffic, traffic_source = None, None   # (roofline.traffic_basis: 'micro' = PMC passes of tools/dcn_micro.py in this run, 'profile record' = the file under profiles/)
    try:
        with open(os.path.join(ROOT, PMC_PROFILE[args.nf])) as f:
            pmc = json.load(f)
        if nl > 0:
            traffic = round(pmc['hbm_bytes_per_pixel'] * (kbytes / nl) / pmc['algorithmic_bytes_per_pixel'])
            traffic_source = '%s (offline PMC pass at offset std %s px, rescaled to this launch size; not measured in this run)' \
                % (PMC_PROFILE[args.nf], pmc['shape'].get('offset_std_px'))
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        l1 = off.get('pcd_align.L1_dcnpack', (None, None))
        lf = "LapPyr(ssim,cb)" if args.lf_mode == 'ssim' else "LapPyr(cb,cb)"
        line = {
            'metric': 'HR frames/sec (fwd+bwd) on %d-frame %dx%d LR windows' % (N, H, W),
            'value': round(world * B * args.steps / dt, 3),
            'unit': 'HR frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            # tensors, accumulation and all non-GEMM math are f32; the GEMM operands are config.gemm
            'dtype': dtype_of(gemm_mode),
            'data': 'synthetic',
            'config': {'workload': 'EDVR%s nf%d, %d-frame %dx%d LR windows, batch %d per GPU, x4 output, VideoSRModel.'
                                   'optimize_parameters: fwd + %s on Y + GWLoss on CbCr + bwd + Adam step'
                                   % ('-M' if args.nf == 64 else '', args.nf, N, H, W, B, lf),
                       'per_gpu_batch': B, 'global_batch': world * B, 'parallelism': 'sequence-dp%d' % world,
                       'gemm': gemm_mode + GEMM_DESC[gemm_mode],
                       'lf_term': 'ssim (restated IQA_pytorch.SSIM, parity unpinned)' if args.lf_mode == 'ssim' else 'cb',
                       'offset_abs_mean_px': None if l1[0] is None else round(l1[0], 4),
                       'offset_abs_max_px': None if l1[1] is None else round(l1[1], 3),
                       'offset_abs_mean_px_per_dcn': {k.split('.')[-1]: round(v[0], 4) for k, v in off.items()},
                       'offset_px_requested': args.offset_px,
                       'loss_last_step': round(float(loss.item()), 6)},
            'roofline': dict(dcn_roofline('dcn_fwd3_kernel (+ its weight pre-pack), fused DCN forward', args.nf, args.nf, gemm_mode, timer),
                             traffic=traffic, traffic_source=traffic_source, traffic_basis=None if traffic is None else 'profile record',

                             dcn_bwd_ms_per_step=round(timer.backward_ms() / max(args.steps, 1), 3)),
        }
        if kms > 0:
            # what the fused DCN forward is actually bound by (profiles/r04_notes.md): not HBM -- its traffic is 1.12x the algorithmic bytes --
            # but what one pixel costs on chip.  The LDS side of that: gather + weight-fragment bytes vs the measured 146 B/clk/CU x 256 CUs x 2.4 GHz;
            # bank conflicts of the per-pixel gather (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.41 at i.i.d. offsets) multiply the LDS
            # cycles by ~1.7, and the SIMDs' instruction issue (412 instructions per four (wave, tap)s, SQ_ACTIVE_INST_ANY 84 %) is the other half
            # peak: the micro-benchmarked ds_read_b128 rate (profiles/r03_lds_micro.txt: 7 clk per 1 KB wave instruction per CU = 146 B/clk)
            # at the 2.4 GHz nominal clock -- round 4 used the 256 B/clk of the CDNA4 data sheet, which no instruction mix here reaches
            lds_peak = 256 * (1024 / 7.0) * 2.4e9 / 1e12
            lds_ach = timer.lds_bytes / (kms * 1e-3) / 1e12
            line['roofline_lds'] = {'kernel': 'dcn_fwd3_kernel', 'bound': 'lds', 'achieved': round(lds_ach, 2), 'peak': round(lds_peak, 1), 'unit': 'TB/s',
                                    'frac': round(lds_ach / lds_peak, 4), 'bytes_per_pixel': round(timer.lds_bytes / max(kbytes, 1) * 4.0 * (args.nf + 216 + args.nf)),
                                    'note': 'conflict-free LDS bytes of the formulation (corner gather + weight fragments) over the timed launches; '
                                            'measured bank conflicts multiply the LDS cycles by ~1.7 (profiles/r03_dcn_sq_counters.json)'}
        if allreduce is not None:
            line['allreduce'] = allreduce
        line['roofline_conv'] = conv_roofline(model.netG, B * N, args.nf, H, W, gemm_mode, sustained=not args.no_sweep)
        if world == 1 and not args.no_sweep:
            nxt = args.warmup + args.steps + 1
            line['offset_sweep'] = offset_sweep(model, x, nxt, native if args.offset_px is not None else None,
                                                pxs=(3.0,) if args.offset_px is not None else (1.0, 3.0))
            if gemm_mode == 'bf16x3':
                line['f32_mode_ms_per_step'] = f32_mode_step(model, nxt)
                # the opt-in speed modes (realvsr_amd.set_gemm_mode): same step, 2 untimed + 3 timed; their parity rows follow below
                line['speed_modes'] = {}
                for m in ('f16fp8', 'bf16x2', 'bf16'):
                    r_m = gemm_mode_step(model, nxt, m, steps=3)
                    line['speed_modes'][m] = dict({'gemm': m + GEMM_DESC[m], 'frames_per_s': round(B * world * 1e3 / r_m['ms_per_step'], 2)}, **r_m)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'], ora = cpu_baseline(args, model.netG.state_dict())
            line['parity'] = parity_check(model, ora)
            line['parity']['gemm'] = gemm_mode
            for m in line.get('speed_modes', {}):   # the same check in the speed modes
                rlib.set_gemm_mode(m)
                try:
                    pm = parity_check(model, ora)
                finally:
                    rlib.set_gemm_mode(gemm_mode)
                line['speed_modes'][m]['parity'] = {k: pm[k] for k in ('out_rel_err', 'residual_rel_err', 'psnr_y_delta_db_at_30db', 'out_max_abs_err',
                                                                        'loss_rel_err', 'grad_l2_err_all', 'worst_param', 'worst_param_l2_err')}
            del ora
        if world == 1 and not args.no_extra and args.config == 2:
            del model, x, gt, timer
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            extra = {}
            try:
                extra['config3'] = extra_train_line(args, 128, 7, 16)
            except Exception as e:   # a side line must not cost the driver its main line
                extra['config3'] = {'error': repr(e)[:300]}
            gc.collect()
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            try:
                extra['config5'] = infer_line(_Cfg(nf=128, nframes=7, height=540, width=960, back_rbs=args.back_rbs), T=10,
                                              offset_px=args.offset_px)
            except Exception as e:
                extra['config5'] = {'error': repr(e)[:300]}
            line['extra'] = extra
        if world == 1 and nl > 0 and not args.no_sweep and H == 180 and W == 320:
            # roofline.traffic measured live when rocprofv3 is on this box (two child processes, ~15 s, as the last thing before the line is printed:
            # every timed measurement of this process ran on a warm GPU); the record kept under profiles/ stays as the fallback
            torch.cuda.empty_cache()
            bpp, note = measure_dcn_traffic(args.nf)
            if bpp is not None:
                line['roofline']['traffic'] = round(bpp * (kbytes / nl) / (4.0 * (args.nf + 216 + args.nf)))
                line['roofline']['traffic_source'] = note
                line['roofline']['traffic_basis'] = 'micro'   # counters of tools/dcn_micro.py (same kernel, L1 shape), rescaled -- not of the timed launches
            elif line['roofline'].get('traffic_source') is not None:
                line['roofline']['traffic_source'] += '; live PMC pass unavailable: ' + note
        print(json.dumps(line), flush=True)
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
