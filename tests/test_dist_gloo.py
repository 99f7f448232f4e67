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


def test_shard_range():
    from realvsr_amd.dist import shard_range
    for total, world in [(128, 8), (10, 4), (3, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
