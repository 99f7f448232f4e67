"""Sequence-data-parallel training support: one process per GPU, gradients averaged with a
bucketed all-reduce over RCCL/xGMI that overlaps the backward pass.

The reference shards the batch the same way (DistributedDataParallel + DistIterSampler,
codes/train.py:19-26, codes/data/__init__.py:10-15, VideoSR_AllPair_model_YCbCr_Split.py:33-34);
N-frame windows are independent, so the only exchange on the hot path is one gradient
all-reduce per step (13.2 MB for EDVR-M).

Design for xGMI (point-to-point links, ring all-reduce is per-link bound): all gradients live in
ONE flat f32 buffer (p.grad are views; the same buffer ``optim.FlatAdam`` consumes), cut into a few
large buckets in reverse registration order (~ the order backward produces them); a bucket's
all-reduce is issued asynchronously the moment its last gradient has been accumulated, so the big
early buckets (reconstruction trunk, fusion) travel while the PCD/feature-extraction backward is
still running.  Collectives are always issued in ascending bucket order on every rank (a bucket that
completes early waits for its predecessors), so ranks whose parameters receive gradients in a
different order -- or not at all -- still pair up the same collectives.
"""
import torch
import torch.distributed as dist

import os

from .optim import FlatBuffers

_DIST_CHECK = os.environ.get('RVSR_DIST_CHECK', '0') == '1'


def broadcast_parameters(module_or_tensors, src=0, process_group=None):
    """Make every rank start from rank `src`'s parameters and buffers (DistributedDataParallel does this in its
    constructor, VideoSR_AllPair_model_YCbCr_Split.py:33-34)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    if isinstance(module_or_tensors, torch.nn.Module):
        tensors = [p.data for p in module_or_tensors.parameters()] + [b.data for b in module_or_tensors.buffers()]
    else:
        tensors = [t.data if isinstance(t, torch.nn.Parameter) else t for t in module_or_tensors]
    for t in tensors:
        dist.broadcast(t, src=src, group=process_group)
    from . import functional as RF
    RF.packed_weights.invalidate()   # parameters were overwritten through .data (no version bump)


class BucketedGradAllReduce:
    def __init__(self, params, bucket_mb=4.0, process_group=None, buffers=None, broadcast=True, force=False):
        """params: iterable of parameters (ignored when `buffers`, a FlatBuffers that already holds them, is given).
        force: run the whole reducer path -- gradient hooks, one asynchronous all_reduce per bucket, finish() -- even in a
        one-rank group, where it is arithmetically the identity (sum over one rank, times 1/1).  That is how a 1-GPU box
        exercises RCCL itself (tests/test_gpu_dist.py, `bench.py --force-allreduce`); needs an initialised process group."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_initialized())
        self.buffers = buffers if buffers is not None else FlatBuffers([list(params)])
        self.params = list(self.buffers.order)
        self.flat = self.buffers.grad
        if broadcast and self.world > 1:
            dist.broadcast(self.buffers.param, src=0, group=process_group)   # one collective for all parameters
            from . import functional as RF
            RF.packed_weights.invalidate()
        self.buckets = []       # (start, end) ranges in self.flat, ascending
        self._bucket_of = {}
        cap = int(bucket_mb * (1 << 20) / 4)
        start = 0
        for i, p in enumerate(self.params):
            self._bucket_of[p] = len(self.buckets)
            end = self.buffers.offset[self.params[i + 1]] if i + 1 < len(self.params) else self.buffers.numel
            if end - start >= cap:
                self.buckets.append((start, end))
                start = end
        if self.buffers.numel > start:
            self.buckets.append((start, self.buffers.numel))
        self._need = [0] * len(self.buckets)
        for p in self.params:
            self._need[self._bucket_of[p]] += 1
        self._pending = list(self._need)
        self._next = 0          # next bucket to issue (strictly ascending on every rank)
        self._works = []
        self.reset_stats()
        if self.active:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def reset_stats(self):
        """Evidence for the scaling run (bench.py `allreduce`): how many buckets left during backward, and how long the
        compute stream waited for the collectives once backward had finished."""
        self.stats_issued_in_backward = 0
        self._exposed = []      # (event at finish() entry, event after the last wait) per step

    def exposed_ms(self):
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    def zero_grad(self):
        self.buffers.zero_grad()
        self._pending = list(self._need)
        self._next = 0
        self._works = []

    def _issue_ready(self, from_hook=False):
        if from_hook and _DIST_CHECK:
            # developer check mode: no collective leaves before finish() has compared the ranks' gradient sets -- ranks that disagree have
            # issued different numbers of bucket collectives by then, and the check's own all_reduce would pair up with one of those
            return
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            s, e = self.buckets[self._next]
            self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._next += 1

    def _hook(self, p):
        self.buffers.rebind(p)   # (a gradient adopted from a plain torch module lives elsewhere: copy it into the bucket)
        self._pending[self._bucket_of[p]] -= 1
        self._issue_ready(from_hook=True)

    def finish(self):
        """Wait for the in-flight buckets and turn the sums into means. Call before optimizer.step()."""
        if not self.active:
            return
        self.stats_issued_in_backward = self._next
        ev = None
        if self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.buffers.rebind()            # (records FlatBuffers.no_grad first: the parameters FlatAdam.step must not move)
        self.buffers.check_bound()
        if _DIST_CHECK:
            # developer check: every rank skipped the same parameters (same autograd graph everywhere)
            mine = torch.zeros(len(self.params), dtype=torch.int32, device=self.flat.device)
            idx = {p: i for i, p in enumerate(self.params)}
            for q in self.buffers.no_grad or []:
                mine[idx[q]] = 1
            lo, hi = mine.clone(), mine.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError('ranks disagree on which parameters received a gradient this step (data-dependent graph?)')
        for b in range(self._next, len(self.buckets)):   # buckets holding parameters that got no gradient this step
            self._pending[b] = 0
        self._issue_ready()
        for w in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
            if len(self._exposed) > 64:
                del self._exposed[:32]
        self.flat.mul_(1.0 / self.world)
        self._works = []


def shard_range(total, rank, world):
    """Contiguous, balanced split of `total` independent windows over `world` ranks."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
