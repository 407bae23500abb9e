"""Data-parallel training over the GPUs of one node: one process per GPU, scenes sharded
by rank, identical replicas, ONE exchange per step -- an all-reduce (RCCL over xGMI when
the backend is "nccl") of the flat fp32 gradient.  The reference has no distributed code
at all; batch-norm statistics stay per replica on purpose
(SURVEY.md section 5): the reference semantics are single-GPU batch statistics.

Gradients live in one contiguous buffer (parameters' .grad are views into it).  The buffer
is cut into a few large contiguous buckets in parameter order; backward produces gradients
in reverse order, so as soon as the FIRST parameter of a bucket has its gradient the whole
bucket is final and its all-reduce is launched asynchronously, overlapping the rest of
backward (the coarse UNet levels and the decoder finish first, the level-0 encoder last).
xGMI is a point-to-point mesh: few, large messages (tens of MB) rather than per-tensor ones.
"""
import torch
import torch.distributed as dist


class FlatGradBucket(object):
    def __init__(self, params, n_buckets=4):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        self.offsets = []
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.offsets.append(off)
            off += p.numel()
        # bucket boundaries at parameter boundaries, roughly equal sizes
        self.buckets = []  # (start, end, index of first parameter)
        target = (n + n_buckets - 1) // n_buckets
        start, first = 0, 0
        for i, p in enumerate(self.params):
            end = self.offsets[i] + p.numel()
            if end - start >= target or i == len(self.params) - 1:
                self.buckets.append((start, end, first))
                start, first = end, i + 1
        self._work = []
        self._hooks = []
        self._launched = set()
        self._first_of = {}
        # this object owns the (zeroed) gradient buffers: let the convolutions accumulate dWeight straight into them
        from .sparseconvnet import conv as _conv
        _conv.ACCUMULATE_INTO_GRAD = True
        _conv.GRAD_WRITTEN = self._grad_written

    def _grad_written(self, param):
        """a convolution accumulated its weight gradient directly (no autograd hook fires for it)"""
        b = self._first_of.get(id(param))
        if b is not None:
            self._launch(b)

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def enable_overlap(self):
        """launch each bucket's all-reduce from a hook on the bucket's first parameter (the last of the
        bucket to receive its gradient in backward)"""
        if self._hooks:
            return
        for b, (start, end, first) in enumerate(self.buckets):
            def hook(_param, b=b):
                self._launch(b)
            self._hooks.append(self.params[first].register_post_accumulate_grad_hook(hook))
            self._first_of[id(self.params[first])] = b

    def _launch(self, b):
        if b in self._launched or not self._distributed():
            return
        self._launched.add(b)
        start, end, _ = self.buckets[b]
        self._work.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, async_op=True))

    def zero(self):
        self.flat.zero_()
        self._work = []
        self._launched = set()

    def all_reduce_mean(self):
        """finish the exchange: launch whatever the hooks have not launched, wait, average.
        No-op in a single-process run."""
        if not self._distributed():
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self._work:
            w.wait()
        self._work = []
        self.flat.div_(dist.get_world_size())


def shard(items, rank, world):
    """scenes of a global batch owned by this rank (contiguous blocks, as a DistributedSampler would)"""
    per = len(items) // world
    assert per * world == len(items), "global batch must divide evenly over ranks"
    return items[rank * per:(rank + 1) * per]
