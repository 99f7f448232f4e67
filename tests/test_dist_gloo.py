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
