"""The N > 1 path on real kernels: two ranks sharing ONE MI355X over gloo (the driver has no multi-GPU box for tests;
RCCL itself is exercised by the driver's scaling run).  -m gpu

  * the real EDVR training step through VideoSRModel(dist) -> FlatAdam buffers -> BucketedGradAllReduce hooks on the
    custom autograd Functions: averaged gradients == full-batch gradients, ranks end with identical parameters
  * `python bench.py --gpus 2` with WORLD_SIZE unset spawns its own ranks and reports n_gpus = 2
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO
from gpu_util import dev

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _opt(dist):
    net = dict(which_model_G='EDVR', nf=16, nc=3, nframes=3, groups=4, front_RBs=1, back_RBs=1, center=None, predeblur=False,
               HR_in=False, w_TSA=True)
    return {'model': 'VideoSR_AllPair_YCbCr_Split', 'dist': dist, 'gpu_ids': [0], 'is_train': True, 'scale': 4, 'augment': None,
            'network_G': net, 'path': {'pretrain_model_G': None, 'strict_load': True},
            'train': {'pixel_criterion_y': 'lappyr', 'pixel_weight_y': 1.0, 'pixel_criterion_c': 'gw', 'pixel_weight_c': 0.5,
                      'weight_decay_G': 0, 'ft_tsa_only': 0, 'lr_G': 1e-3, 'beta1': 0.9, 'beta2': 0.99, 'bucket_mb': 0.05}}


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.rand(4, 3, 3, 24, 32, generator=g), torch.rand(4, 3, 96, 128, generator=g)


def _build(dist, seed):
    sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
    from weights import fill_state_dict
    from realvsr_amd.VideoSR_model import create_model
    torch.manual_seed(seed)
    model = create_model(_opt(dist))
    return model, fill_state_dict


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from realvsr_amd.dist import shard_range, broadcast_parameters
    model, fill = _build(True, seed=100 + rank)          # different initial weights per rank ...
    if rank == 0:
        fill(model.netG, 808, offset_std=0.02)
    broadcast_parameters(model.netG)                     # ... until rank 0's are broadcast
    assert len(model.reducer.buckets) > 3
    x, gt = _data()
    s, e = shard_range(4, rank, world)
    model.feed_data({'LQs': x[s:e], 'GT': gt[s:e]})
    for step in (1, 2):
        model.optimize_parameters(step)
        if step == 1:
            grads = model.optimizer_G.buffers.grad.detach().cpu().clone()
    # numpy payloads: torch tensors travel through an fd-sharing side channel that dies with this process
    q.put((rank, grads.numpy(), model.optimizer_G.buffers.param.detach().cpu().numpy(), model.get_current_log()['l_pix']))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_match_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, grads, params, l = q.get(timeout=600)
        got[r] = (torch.from_numpy(grads), torch.from_numpy(params), l)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process, full batch of 4
    torch.cuda.set_device(0)
    model, fill = _build(False, seed=7)
    fill(model.netG, 808, offset_std=0.02)
    x, gt = _data()
    model.feed_data({'LQs': x, 'GT': gt})
    model.optimize_parameters(1)
    ref_g = model.optimizer_G.buffers.grad.detach().cpu().clone()
    model.optimize_parameters(2)
    ref_p = model.optimizer_G.buffers.param.detach().cpu()
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])      # ranks agree bit for bit
    err = ((got[0][0].double() - ref_g.double()).norm() / ref_g.double().norm()).item()
    print('averaged gradient vs full-batch gradient: rel l2 err %.3e' % err)
    assert err <= 2e-4, err
    # two Adam steps from the same start: updates agree
    perr = ((got[0][1].double() - ref_p.double()).norm() / ref_p.double().norm()).item()
    print('parameters after 2 steps, 2 ranks vs 1: rel l2 err %.3e' % perr)
    assert perr <= 1e-4, perr


def test_bench_self_launches_n_ranks():
    env = dict(os.environ, RVSR_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1',
                          '--height', '32', '--width', '48', '--no-cpu-baseline', '--no-extra'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 2 and rec['config']['parallelism'] == 'sequence-dp2'
    assert rec['value'] > 0 and rec['roofline']['launches'] == 2 * 4
    # evidence for the day the scaling run happens: what was all-reduced, how much of it during backward, what stayed exposed
    ar = rec['allreduce']
    assert ar['buckets'] >= 1 and ar['bytes'] == sum(ar['bucket_bytes']) and ar['params_identical_after_last_step'] is True
    assert 0 <= ar['issued_during_backward'] <= ar['buckets'] and ar['exposed_ms'] >= 0.0


def test_bench_line_carries_offset_sweep_and_f32_step():
    """The driver-timed line (N = 1) runs at 1 px mean offsets and reports the raw-init and large-motion steps, the exact-f32 step, the
    LDS roofline object and the config-3 / config-5 side lines next to the headline."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', '1',
                          '--height', '32', '--width', '48', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert abs(rec['config']['offset_abs_mean_px'] - 1.0) < 0.2 and rec['config']['offset_px_requested'] == 1.0
    assert set(rec['offset_sweep']) == {'raw_init', '3px'}
    for k, v in rec['offset_sweep'].items():
        assert v['ms_per_step'] > 0 and v['dcn_bwd_ms'] > 0 and 0 < v['dcn_fwd_frac'] < 1
        if k == 'raw_init':
            assert v['offset_abs_mean_px'] < 0.1
        else:
            assert abs(v['offset_abs_mean_px'] - float(k[:-2])) < 0.2 * float(k[:-2])
    assert 0 < rec['roofline_lds']['frac'] < 1 and rec['roofline_lds']['bound'] == 'lds'
    assert rec['extra']['config3']['ms_per_step'] > 0 and rec['extra']['config5']['ms_per_frame'] > 0 and rec['extra']['config5']['graph_bit_identical']
    assert rec['f32_mode_ms_per_step'] > 0 and rec['roofline']['dcn_bwd_ms_per_step'] > 0
    assert 'allreduce' not in rec
