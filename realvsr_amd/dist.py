"""Sequence-data-parallel training support: one process per GPU, gradients averaged with a
bucketed all-reduce over RCCL/xGMI that overlaps the backward pass.

The reference shards the batch the same way (DistributedDataParallel + DistIterSampler,
codes/train.py:19-26, codes/data/__init__.py:10-15, VideoSR_AllPair_model_YCbCr_Split.py:33-34);
N-frame windows are independent, so the only exchange on the hot path is one gradient
all-reduce per step (13.2 MB for EDVR-M).

Design for xGMI (point-to-point links, ring all-reduce is per-link bound): all gradients live in
ONE flat f32 buffer (p.grad are views), cut into a few large buckets in reverse registration
order (~ the order backward produces them); a bucket's all-reduce is issued asynchronously the
moment its last gradient has been accumulated, so the big early buckets (reconstruction trunk,
fusion) travel while the PCD/feature-extraction backward is still running.
"""
import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    def __init__(self, params, bucket_mb=4.0, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        # reverse order: parameters used last in forward get their gradients first
        order = list(reversed(self.params))
        self.buckets = []       # (start, end) ranges in self.flat
        self._bucket_of = {}
        off, start, cap = 0, 0, int(bucket_mb * (1 << 20) / 4)
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            if off - start >= cap:
                self.buckets.append((start, off))
                start = off
        if off > start:
            self.buckets.append((start, off))
        self._need = [0] * len(self.buckets)
        for p in order:
            self._need[self._bucket_of[p]] += 1
        self._pending = list(self._need)
        self._works = []
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def zero_grad(self):
        self.flat.zero_()
        self._pending = list(self._need)
        self._works = []

    def _hook(self, p):
        b = self._bucket_of[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            s, e = self.buckets[b]
            self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the in-flight buckets and turn the sums into means. Call before optimizer.step()."""
        if self.world == 1:
            return
        for b, left in enumerate(self._pending):  # parameters that received no gradient this step
            if left > 0:
                s, e = self.buckets[b]
                self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self.flat.mul_(1.0 / self.world)
        self._works = []


def shard_range(total, rank, world):
    """Contiguous, balanced split of `total` independent windows over `world` ranks."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
