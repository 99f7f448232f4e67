"""world_size-2 gloo test of the bucketed gradient all-reduce (host logic; CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, 1, 1), nn.ReLU(), nn.Conv2d(8, 8, 3, 1, 1), nn.ReLU(), nn.Conv2d(8, 3, 3, 1, 1))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from realvsr_amd.dist import BucketedGradAllReduce, shard_range
    net = _toy()
    red = BucketedGradAllReduce(net.parameters(), bucket_mb=0.001)  # tiny buckets -> several in flight
    assert len(red.buckets) > 1
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(1))
    s, e = shard_range(4, rank, world)
    for _ in range(2):  # twice: zero_grad / hook state must reset correctly
        red.zero_grad()
        net(x[s:e]).square().mean().backward()
        red.finish()
    q.put((rank, torch.cat([p.grad.flatten() for p in net.parameters()]).clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    net = _toy()
    x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(1))
    net(x).square().mean().backward()  # mean over the full batch == mean of the per-rank means
    ref = torch.cat([p.grad.flatten() for p in net.parameters()])
    for r in range(world):
        assert torch.allclose(got[r], ref, rtol=1e-5, atol=1e-7), r
    assert torch.equal(got[0], got[1])


def _skip_worker(rank, world, port, q):
    """FlatAdam behind the bucketed all-reduce: a parameter that gets no gradient must not move on ANY rank (ADVICE r3:
    finish() rebinds every parameter before step() could see `grad is None`)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RVSR_DIST_CHECK='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from realvsr_amd import optim, functional as RF
    import realvsr_amd.dist as rdist
    rdist._DIST_CHECK = True

    def adam_cpu(param, grad, m, v, step_size, b1, b2, eps, wd, bc2s):   # the kernel needs a GPU; same arithmetic in torch
        g = grad + wd * param if wd else grad
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        param.addcdiv_(m, v.sqrt() / bc2s + eps, value=-step_size)

    RF.adam_step_ = adam_cpu
    torch.manual_seed(0)
    a, b = torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(7))
    opt = optim.FlatAdam([a, b], lr=1e-2, weight_decay=0.1)
    red = rdist.BucketedGradAllReduce(None, bucket_mb=0.00001, buffers=opt.buffers, broadcast=True)
    hist = []
    for it in range(3):
        red.zero_grad()
        ((a * a).sum() * (rank + 1)).backward()
        if it != 1:
            (b * 3).sum().backward()        # step 1: b receives no gradient on any rank
        b_before, m_before = b.detach().clone(), opt.state[b]['exp_avg'].clone()
        red.finish()
        opt.step()
        if it == 1:
            assert torch.equal(b.detach(), b_before), 'a parameter without gradient moved (weight decay / moment decay)'
            assert torch.equal(opt.state[b]['exp_avg'], m_before)
        else:
            assert not torch.equal(b.detach(), b_before)
        hist.append(torch.cat([a.detach().flatten(), b.detach().flatten()]).clone())
    q.put((rank, torch.stack(hist).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_adam_skip_set_with_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_skip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: torch.from_numpy(v) for r, v in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(got[0], got[1])          # ranks stay bit-identical
    # single-process reference: torch.optim.Adam on the averaged gradient ((1 + 2) / 2 = 1.5 x the rank-0 loss for a)
    torch.manual_seed(0)
    ra, rb = torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(7))
    ref = torch.optim.Adam([ra, rb], lr=1e-2, weight_decay=0.1)
    for it in range(2):                          # (steps 0 and 1: afterwards torch's per-parameter step count lags, documented)
        ref.zero_grad(set_to_none=True)
        ((ra * ra).sum() * 1.5).backward()
        if it != 1:
            (rb * 3).sum().backward()
        ref.step()
        want = torch.cat([ra.detach().flatten(), rb.detach().flatten()])
        assert torch.allclose(got[0][it], want, atol=1e-6), it


def test_shard_range():
    from realvsr_amd.dist import shard_range
    for total, world in [(128, 8), (10, 4), (3, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


# ---- eight-rank rehearsal of BASELINE config 4 on CPU (round 6): the real config-3 architecture's parameter table (nf128, 7 frames, TSA:
# 47.4 MB of gradients, 4 MB buckets), 8 gloo ranks, gradients arriving in a different order on every rank, and one rank whose non-TSA
# parameters receive no gradient at all (the reference's `ft_tsa_only` freeze, VideoSR_AllPair_model_YCbCr_Split.py:103-116, applied on ONE
# rank only -- harsher than anything the reference does): the collectives must still pair up (ascending bucket order on every rank), the
# averages must be exact, the broadcast must leave every rank with rank 0's parameters, and RVSR_DIST_CHECK must name the disagreement.
def _c4_worker(rank, world, port, q, frozen_rank, check):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    torch.set_num_threads(1)
    import realvsr_amd.dist as rdist
    from realvsr_amd.archs.EDVR_arch import EDVR
    from realvsr_amd.optim import FlatBuffers
    rdist._DIST_CHECK = check
    torch.manual_seed(100 + rank)                    # every rank starts from DIFFERENT weights: the reducer broadcasts rank 0's
    net = EDVR(nf=128, nc=3, nframes=7, groups=8, front_RBs=5, back_RBs=10, w_TSA=True)
    names = {p: n for n, p in net.named_parameters()}
    buffers = FlatBuffers([list(net.parameters())])
    red = rdist.BucketedGradAllReduce(None, bucket_mb=4.0, buffers=buffers, broadcast=True)
    psum = buffers.param.double().sum()
    lo, hi = psum.clone(), psum.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    params = list(red.params)
    order = torch.randperm(len(params), generator=torch.Generator().manual_seed(rank)).tolist()   # per-rank gradient arrival order
    err = None
    issued = []
    for step in range(2):
        red.zero_grad()
        loss = 0.0
        for i in order:                               # autograd runs the LAST-created node first: arrival order = reversed(order)
            p = params[i]
            if rank == frozen_rank and not names[p].startswith('tsa_fusion'):
                continue
            loss = loss + (p * ((rank + 1) * (i + 1) * 1e-3)).sum()
        loss.backward()
        try:
            red.finish()
        except RuntimeError as e:                     # (RVSR_DIST_CHECK: every rank raises after the same two collectives)
            err = str(e)
            break
        issued.append(red.stats_issued_in_backward)
    worst = 0.0
    if err is None:
        for i, p in enumerate(params):
            ranks = [r for r in range(world) if not (r == frozen_rank and not names[p].startswith('tsa_fusion'))]
            want = sum((r + 1) for r in ranks) * (i + 1) * 1e-3 / world
            worst = max(worst, float((buffers.grad_view(p) - want).abs().max()) / want)
    q.put((rank, dict(buckets=len(red.buckets), bucket_bytes=[4 * (e - s) for s, e in red.buckets], grad_bytes=4 * buffers.numel,
                      same_params=bool(lo.item() == hi.item()), worst=worst, err=err, issued=issued,
                      no_grad=len(buffers.no_grad or []))))
    if err is not None:
        # after the check has fired the ranks have issued different numbers of bucket collectives: the group cannot be used (or torn
        # down collectively) any more -- the training run is over at that point, and so is this worker
        q.close()
        q.join_thread()
        os._exit(0)
    dist.barrier()
    dist.destroy_process_group()


def _run_c4(frozen_rank, check):
    world, port = 8, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q, frozen_rank, check)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got


def test_config4_eight_ranks_uneven_arrival_and_one_frozen_rank():
    got = _run_c4(frozen_rank=5, check=False)
    r0 = got[0]
    assert r0['grad_bytes'] == 47378176 and r0['buckets'] == 11            # the table `bench.py --config 3 --gpus 8 --dry-run` prints
    assert all(b >= 4 << 20 for b in r0['bucket_bytes'][:-1])
    for r in range(8):
        g = got[r]
        assert g['err'] is None and g['same_params'], (r, g)
        assert g['bucket_bytes'] == r0['bucket_bytes']
        assert g['worst'] < 1e-6, (r, g['worst'])                          # exact means, incl. the buckets rank 5 never filled
        assert all(0 <= n <= 11 for n in g['issued'])
    assert got[5]['issued'] == [0, 0] or max(got[5]['issued']) < 11         # the frozen rank issues most buckets from finish()
    assert got[5]['no_grad'] > 0 and got[0]['no_grad'] == 0


def test_config4_dist_check_names_a_rank_that_skipped_parameters():
    got = _run_c4(frozen_rank=5, check=True)
    for r in range(8):
        assert got[r]['err'] is not None and 'disagree' in got[r]['err'], (r, got[r])
