"""Value type and small helpers of the operator API.

Mirrors sparseconvnet/sparseConvNetTensor.py:13-26, sparseconvnet/metadata.py:16-17 and
sparseconvnet/utils.py:13-28 (behaviour, not text).
"""
import torch

from .. import SCN


def Metadata(dim):
    """factory used by InputLayer (sparseconvnet/metadata.py:16-17)"""
    if dim != 3:
        raise NotImplementedError("only dimension 3 is reachable in the reference GPU path "
                                  "(sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cpp:40-44)")
    return SCN.Metadata_3()


class GradJoin(object):
    """Two autograd Functions whose input gradients meet at ONE tensor x (the shortcut and the body of a residual block)
    share one of these, created in forward.  In backward the first to finish leaves its gradient here and returns None
    to autograd; the second adds it INSIDE its own kernel (batch norm backward's apply pass, the NiN dInput's output
    tile) and returns the sum: the elementwise add kernel autograd would launch at x disappears.  Whichever order the
    engine runs the two in, the sum is the same.

    What is left here is stamped with the backward pass (graph task) and the SIDE that left it, because one object lives as
    long as its forward pass' graph and backward passes can fail, be repeated over a retained graph, or nest:
      * same pass, other side                         -> taken (the normal case);
      * same pass, same side                          -> RuntimeError (a gradient would be lost);
      * another pass that COMPLETED, other side       -> taken: re-entrant backward (checkpointing) ran the first side in a
        nested pass of its own (round 4 dropped this one silently: a summand of dL/dx was lost -- round-4 advisor);
      * another pass, same side                       -> stale (that pass never ran the other side, or failed): replaced;
      * another pass that has NOT completed, other side -> cannot be told apart: a failed pass being repeated over a retained
        graph (stale: must be dropped) or an enclosing pass that is still running (must be added).  RuntimeError, never a
        silently wrong gradient; `containers.GRAD_JOIN = False` avoids the situation."""
    __slots__ = ("pending", "armed", "left", "task", "side", "done", "token")

    def __init__(self, armed=True):
        self.pending = None
        self.task = None
        self.side = None
        self.done = False   # the pass that left `pending` has completed (its final callbacks ran)
        self.token = None   # identity of the last put (the completion callback must not hold the gradient itself)
        self.armed = armed  # False: created by one side only (JoinConcatTable); the other side (JoinTable) arms it in forward
        self.left = None    # JoinConcatTable: the features tensor whose two gradients meet here (JoinTable checks it is its left entry)

    @staticmethod
    def _task():
        try:  # the backward pass that is running (a retried or second backward over a retained graph is another one)
            return torch._C._current_graph_task_id()
        except AttributeError:
            return None

    def put(self, grad, side=None):
        task = self._task()
        if self.pending is not None and self.task == task:
            self.pending = None
            raise RuntimeError("occuseg_amd GradJoin: two gradients left in one backward pass (by %r and %r) and nobody took "
                               "the first" % (self.side, side))
        # (whatever an EARLIER pass left is stale by now: this side would have taken it otherwise -- see take)
        tok = object()
        self.pending, self.task, self.side, self.done, self.token = grad, task, side, False, tok
        try:  # final callbacks run when the pass completes; a pass that raises skips them
            # (the callback names the put by a token, not by the tensor: a closure over `grad` kept every parked gradient --
            #  ~150 MB each at level 0 of the m64 network -- alive until the whole backward pass had ended; round-5 advisor)
            def _completed(join=self, task=task, tok=tok):
                if join.token is tok and join.pending is not None and join.task == task:
                    join.done = True
            torch.autograd.Variable._execution_engine.queue_callback(_completed)
        except RuntimeError:  # (not inside a backward pass: the boundary functions called by hand)
            self.done = True

    def take(self, side=None):
        g = self.pending
        if g is None:
            return None
        same_side = side is not None and side == self.side
        if self.task == self._task():
            if same_side:
                self.pending = None
                raise RuntimeError("occuseg_amd GradJoin: side %r came twice in one backward pass" % (side,))
            self.pending = None
            return g
        if same_side:      # left by this very side in an earlier pass that never ran the other side: stale
            self.pending = None
            return None
        if self.done:      # left by a pass that completed: a nested (re-entrant) pass ran the other side
            self.pending = None
            return g
        self.pending = None
        raise RuntimeError(
            "occuseg_amd GradJoin: side %r finds a gradient left by side %r in ANOTHER backward pass that has not completed -- "
            "either a failed pass is being repeated over a retained graph (the gradient is stale) or an enclosing pass is still "
            "running (re-entrant checkpointing; the gradient belongs to this sum).  The two cannot be told apart, so nothing is "
            "added or dropped silently: run the forward pass again, or set occuseg_amd.sparseconvnet.containers.GRAD_JOIN = False"
            % (side, self.side))


class SparseConvNetTensor(object):
    """features [N, planes] + the shared Metadata + the spatial size of the grid."""

    def __init__(self, features=None, metadata=None, spatial_size=None):
        self._features = features
        self.metadata = metadata
        self.spatial_size = spatial_size
        self._colstats = None  # (version of the features tensor, [(planes, partial column sums), ...]) -- see set_colstats
        self._bn_link = None   # (version, SCN.BnLink) left by the batch norm that produced `features` -- see set_bn_link
        self.join_base = None  # [rows, 2 planes] buffer whose right half `features` is (a Deconvolution with join_right)
        self.left_join = None  # GradJoin offered by a JoinConcatTable for the gradient of the OTHER table entry

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, value):
        # whatever statistics the producing kernel handed over belong to the tensor that was here before
        self._features = value
        self._colstats = None
        self._bn_link = None
        self.join_base = None
        self.left_join = None

    def set_colstats(self, segments):
        """Column statistics of ``features`` emitted by the kernel(s) that produced it (not in the reference).  They
        travel with THIS assignment of ``features``: assigning ``features`` again drops them, and an in-place
        operation on the tensor (version counter) invalidates them.  Not visible to either: a write through
        ``features.data`` -- call ``set_colstats(None)`` after one (the batch norm then reads the tensor itself)."""
        f = self._features
        self._colstats = (f._version, list(segments)) if segments and f is not None else None

    def get_colstats(self):
        if self._colstats is None or self._features is None:
            return None
        return self._colstats[1] if self._colstats[0] == self._features._version else None

    def set_bn_link(self, link):
        """``features`` is the output of a batch norm: the convolution that consumes it hands that batch norm's backward
        pass its two reductions (SCN.BnLink).  Same validity rules as set_colstats."""
        self._bn_link = (self._features._version, link) if link is not None and self._features is not None else None

    def get_bn_link(self):
        if self._bn_link is None or self._features is None:
            return None
        return self._bn_link[1] if self._bn_link[0] == self._features._version else None

    def get_spatial_locations(self, spatial_size=None):
        """LongTensor [N,4]: coordinates and batch index of the active sites"""
        size = self.spatial_size if spatial_size is None else spatial_size
        return self.metadata.getSpatialLocations(size)

    def type(self, t=None):
        if t is None:
            return self.features.type()
        self.features = self.features.type(t)
        return self

    def cuda(self):
        self.features = self.features.cuda()
        return self

    def cpu(self):
        self.features = self.features.cpu()
        return self

    def __repr__(self):
        return "SparseConvNetTensor<<features.shape=%r, spatial size=%r>>" % (
            None if self.features is None else tuple(self.features.shape), self.spatial_size)


def toLongTensor(dimension, x):
    """int | list | LongTensor -> LongTensor[dimension] (sparseconvnet/utils.py:13-20)"""
    if torch.is_tensor(x) and x.dtype == torch.int64 and not x.is_cuda:
        return x
    if isinstance(x, (list, tuple)):
        assert len(x) == dimension
        return torch.tensor(list(x), dtype=torch.int64)
    return torch.full((dimension,), int(x), dtype=torch.int64)


def optionalTensor(module, name):
    """missing parameter -> zero-element tensor, read as NULL by the backend
    (sparseconvnet/utils.py:23-24, sparseconvnet/SCN/Metadata/Metadata.cpp:28-30)"""
    t = getattr(module, name, None)
    return t if t is not None else torch.Tensor()


def optionalTensorReturn(t):
    return t if t.numel() else None


def like_empty(t):
    """the `input_features.new()` of the reference: an empty tensor on the same device"""
    return torch.empty(0, dtype=t.dtype, device=t.device)



class _NoGraphCtx(object):
    """what a Function's forward sees of autograd's context when no graph is recorded"""
    needs_input_grad = ()

    def save_for_backward(self, *tensors):
        pass

    def mark_dirty(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


def apply_function(fn, *args):
    """``fn.apply(*args)``; under torch.no_grad() the forward is called directly -- Function.apply costs ~6 us of host time
    per call whether or not it records anything, and the forward-only configuration is bound by its host time (80 calls per
    forward pass of a UNet)."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoGraphCtx(), *args)
