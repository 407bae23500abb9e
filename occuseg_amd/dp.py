"""Data-parallel training over the GPUs of one node: one process per GPU, scenes sharded
by rank, identical replicas, ONE exchange per step -- an all-reduce (RCCL over xGMI when
the backend is "nccl") of the flat fp32 gradient.  The reference has no distributed code
at all; batch-norm statistics stay per replica on purpose
(SURVEY.md section 5): the reference semantics are single-GPU batch statistics.

Gradients live in one contiguous buffer (parameters' .grad are views into it), so the
collective is a single large message: on a point-to-point xGMI mesh a few big transfers
beat many small ones.
"""
import torch
import torch.distributed as dist


class FlatGradBucket(object):
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self):
        """average the gradient over ranks; no-op in a single-process run"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def shard(items, rank, world):
    """scenes of a global batch owned by this rank (contiguous blocks, as a DistributedSampler would)"""
    per = len(items) // world
    assert per * world == len(items), "global batch must divide evenly over ranks"
    return items[rank * per:(rank + 1) * per]
