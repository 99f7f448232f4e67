#!/usr/bin/env python
"""bench.py -- HR frames/sec (fwd + loss + bwd + optimizer step) of the EDVR hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run,
one rank per GPU over RCCL).  W untimed warm-up steps, then EXACTLY K steps bracketed by
barrier + torch.cuda.synchronize() on both sides; max over ranks; rank 0 prints ONE JSON line.

Workload = BASELINE.json configs[1] ("EDVR-M 64ch, 5-frame 180x320 LR, batch 8, fwd+bwd on
1xMI355X"): EDVR(nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10, w_TSA=True), x4 output,
loss = LapPyrLoss(3,'cb','cb') on Y + GWLoss(w=4) on CbCr (the composition of the reference's
optimize_parameters, VideoSR_AllPair_model_YCbCr_Split.py:163-191, with the in-tree 'cb' low-frequency term), Adam.  Synthetic data:
x ~ U[0,1) seed 1234, GT ~ U[0,1) seed 1235, default module init under seed 0 with
conv_offset_mask.weight ~ N(0, 0.01^2) so the deformable offsets are non-zero.  Weak scaling:
every rank processes its own B=8 windows; value = N*B*K / time.

Extra objects on the line:
  roofline     -- the DCN forward kernel (dcn_fwd_kernel, the kernel north_star grades): algorithmic
                  bytes 4*(C+216+Co) per output pixel summed over the timed DCN
                  launches / their HIP-event durations, vs the 8 TB/s HBM3E peak.
  cpu_baseline -- the CPU oracle (oracle/edvr_oracle.py, kind "port") on ONE window of the same
                  workload (B=1), timed on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', str(min(os.cpu_count() or 1, 32)))  # CPU oracle threads (cpu_baseline)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_net(nf, nframes, back_RBs, device):
    from realvsr_amd.archs.EDVR_arch import EDVR
    torch.manual_seed(0)
    net = EDVR(nf=nf, nc=3, nframes=nframes, groups=8, front_RBs=5, back_RBs=back_RBs, w_TSA=True)
    gen = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if 'conv_offset_mask.weight' in name:
                p.copy_(torch.randn(p.shape, generator=gen) * 0.01)
    return net.to(device)


def make_batch(B, N, H, W, device, rank=0):
    x = torch.rand(B, N, 3, H, W, generator=torch.Generator().manual_seed(1234 + 1000 * rank))
    gt = torch.rand(B, 3, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235 + 1000 * rank))
    return x.to(device), gt.to(device)


class DcnTimer:
    """HIP events around every fused-DCN forward launch on the launching (current) stream."""

    def __init__(self):
        self.events, self.bytes = [], 0.0

    def install(self):
        from realvsr_amd import functional as RF
        L = RF._lib.lib()
        orig = L.rvsr_dcn_pack_forward
        timer = self

        def timed(*a):
            # a: input, weight, bias, om, output, B, C, H, W, Co, stride, pad, dil, dg, act, slope, stream
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = orig(*a)
            e.record()
            B, C, H, W, Co, stride, pad, dil = a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]
            Ho = (H + 2 * pad - (2 * dil + 1)) // stride + 1
            Wo = (W + 2 * pad - (2 * dil + 1)) // stride + 1
            timer.events.append((s, e))
            timer.bytes += 4.0 * (C * H * W + (216 + Co) * Ho * Wo) * B + 4.0 * Co * C * 9
            return rc

        self._orig, self._L = orig, L
        _Proxy.wrap(RF, 'rvsr_dcn_pack_forward', timed)

    def uninstall(self):
        from realvsr_amd import functional as RF
        _Proxy.unwrap(RF)

    def result(self):
        ms = sum(s.elapsed_time(e) for s, e in self.events)
        n = len(self.events)
        return n, ms, self.bytes


class _Proxy:
    """Minimal proxy so one C-ABI symbol can be intercepted without touching the library object."""

    def __init__(self, lib, name, fn):
        self.__dict__['_lib'], self.__dict__['_name'], self.__dict__['_fn'] = lib, name, fn

    def __getattr__(self, k):
        return self._fn if k == self._name else getattr(self._lib, k)

    @staticmethod
    def wrap(RF, name, fn):
        real = RF._lib.lib()
        RF._lib._lib = _Proxy(real, name, fn)
        _Proxy._real = real

    @staticmethod
    def unwrap(RF):
        RF._lib._lib = _Proxy._real


def cpu_baseline(nf, nframes, back_RBs, H, W):
    """One window (B=1) of the same workload through the CPU oracle: fwd + loss + bwd."""
    from oracle import edvr_oracle as O
    from realvsr_amd.archs.EDVR_arch import EDVR
    torch.manual_seed(0)
    net = EDVR(nf=nf, nc=3, nframes=nframes, groups=8, front_RBs=5, back_RBs=back_RBs, w_TSA=True)
    gen = torch.Generator().manual_seed(99)
    sd = {}
    for k, v in net.state_dict().items():
        v = v.detach().clone()
        if 'conv_offset_mask.weight' in k:
            v = torch.randn(v.shape, generator=gen) * 0.01
        sd[k] = v.requires_grad_(True)
    # bounded thread count: the conv-heavy torch CPU path stops scaling (and collapses from
    # oversubscription) well below the 256 hardware threads of the GPU box's host
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    x = torch.rand(1, nframes, 3, H, W, generator=torch.Generator().manual_seed(1234))
    gt = torch.rand(1, 3, 4 * H, 4 * W, generator=torch.Generator().manual_seed(1235))
    t0 = time.perf_counter()
    out = O.edvr_forward(sd, x, nframes=nframes, groups=8, front_RBs=5, back_RBs=back_RBs, w_TSA=True)
    loss = O.lap_pyr_loss(out[:, 0:1], gt[:, 0:1], 3) + O.gw_loss(out[:, 1:3], gt[:, 1:3], 4)
    loss.backward()
    dt = time.perf_counter() - t0
    return {'value': round(1.0 / dt, 5), 'unit': 'HR frames/s', 'cores': cores, 'kind': 'port',
            'sample': '1 window (B=1, %d frames %dx%d LR) fwd+loss+bwd through oracle/edvr_oracle.py '
                      '(torch %s CPU ops + OpenMP C DCN), %.1f s' % (nframes, H, W, torch.__version__, dt)}


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
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    ndev = torch.cuda.device_count()
    backend = os.environ.get('RVSR_BENCH_BACKEND', 'nccl')  # 'gloo': developer check of the N>1 path on one GPU
    if backend == 'nccl' and world > ndev:
        raise SystemExit('WORLD_SIZE=%d but only %d GPUs are visible (one rank per GPU)' % (world, ndev))
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from realvsr_amd import loss as L
    from realvsr_amd import _lib as rlib
    from realvsr_amd.dist import BucketedGradAllReduce
    gemm_mode = rlib.get_gemm_mode()
    B, N, H, W = args.batch, args.nframes, args.height, args.width
    net = build_net(args.nf, N, args.back_rbs, device)
    x, gt = make_batch(B, N, H, W, device, rank)
    crit_y, crit_c = L.LapPyrLoss(3, 'cb', 'cb', 'mean'), L.GWLoss(w=4)
    reducer = BucketedGradAllReduce(net.parameters(), bucket_mb=4.0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.99))

    def step():
        reducer.zero_grad()
        out = net(x)
        loss = crit_y(out[:, 0:1], gt[:, 0:1]) + crit_c(out[:, 1:3], gt[:, 1:3])
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = DcnTimer()
    timer.install()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    timer.uninstall()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    nl, kms, kbytes = timer.result()
    # HBM traffic of the DCN forward kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 runs,
    # profiles/r01_dcn_fwd_pmc.json): measured bytes per output pixel x the pixels an average timed launch covers
    traffic = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_dcn_fwd_pmc.json')) as f:
            pmc = json.load(f)
        if args.nf == 64 and nl > 0:
            traffic = round(pmc['hbm_bytes_per_pixel'] * (kbytes / nl) / pmc['algorithmic_bytes_per_pixel'])
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        line = {
            'metric': 'HR frames/sec (fwd+bwd) on 5-frame 180x320 LR windows',
            'value': round(world * B * args.steps / dt, 3),
            'unit': 'HR frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',  # tensors, accumulation and all non-GEMM math are f32; see config.gemm for the GEMM operands
            'data': 'synthetic',
            'config': {'workload': 'EDVR-M nf%d, %d-frame %dx%d LR windows, batch %d per GPU, x4 output, '
                                   'fwd + LapPyr(cb,cb) on Y + GWLoss on CbCr + bwd + Adam step'
                                   % (args.nf, N, H, W, B),
                       'per_gpu_batch': B, 'global_batch': world * B, 'parallelism': 'sequence-dp%d' % world,
                       'gemm': gemm_mode + (' (3-term bf16 split on v_mfma_f32_32x32x16_bf16, f32 accumulate)'
                                            if gemm_mode == 'bf16x3' else ' (v_mfma_f32_32x32x2_f32, exact f32)'),
                       'loss_last_step': round(float(loss.item()), 6)},
            'roofline': {'kernel': 'dcn_fwd2_kernel (+ its weight pre-pack), fused DCN forward', 'bound': 'hbm',
                         'achieved': round(kbytes / (kms * 1e-3) / 1e9, 2) if kms > 0 else None,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(kbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kms > 0 else None,
                         'traffic': traffic, 'launches': nl, 'avg_launch_ms': round(kms / max(nl, 1), 4),
                         'algorithmic_bytes_per_launch': round(kbytes / max(nl, 1))},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.nf, N, args.back_rbs, H, W)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
