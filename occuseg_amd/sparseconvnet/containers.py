"""Graph glue: Sequential, ConcatTable, JoinTable, AddTable, Identity, NetworkInNetwork.

Operator API of sparseconvnet/sequential.py:9-17, sparseconvnet/tables.py:13-55, sparseconvnet/identity.py,
sparseconvnet/networkInNetwork.py:14-88.  Children are registered under '0','1',... so that
state_dict keys equal the reference's.
"""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from .. import SCN
from . import counters
from .tensor import GradJoin, SparseConvNetTensor, apply_function, like_empty, optionalTensor


def _scoped(fn):
    """bracket a container's forward with SCN.forward_enter / forward_exit: the outermost container of a forward sweep starts
    a new weight epoch (SCN.py, "Forward scopes")"""
    def forward(self, *args, **kwargs):
        SCN.forward_enter()
        try:
            return fn(self, *args, **kwargs)
        finally:
            SCN.forward_exit()
    forward.__doc__ = fn.__doc__
    return forward


class Sequential(torch.nn.Sequential):
    @_scoped
    def forward(self, input):
        for module in self._modules.values():
            input = module(input)
        return input

    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def input_spatial_size(self, out_size):
        for name in reversed(self._modules):
            out_size = self._modules[name].input_spatial_size(out_size)
        return out_size


class Identity(Module):
    def forward(self, input):
        return input

    def input_spatial_size(self, out_size):
        return out_size


GRAD_JOIN = True  # A/B switch: residual blocks add the two gradients of their input inside a kernel (tensor.GradJoin)


def _like(first, features):
    return SparseConvNetTensor(features, first.metadata, first.spatial_size)


class ConcatTable(Sequential):
    """apply every child to the same input, return the list of results"""

    @_scoped
    def forward(self, input):
        return [child(input) for child in self._modules.values()]

    def input_spatial_size(self, out_size):
        return self._modules["0"].input_spatial_size(out_size)


class ResidualConcatTable(ConcatTable):
    """ConcatTable(shortcut, body) of a residual block (sparseconvnet/networkArchitectures.py:222-229) whose body ends
    in a SubmanifoldConvolution: same children, same state_dict keys, same result as ConcatTable -> AddTable, but the
    shortcut is handed to that last convolution, which adds it inside its kernel (the output tile starts from the
    shortcut's rows).  Returns a one-element list, so the AddTable that follows has nothing left to add."""

    @_scoped
    def forward(self, input):
        from .conv import SubmanifoldConvolution
        children = list(self._modules.values())
        body = list(children[1]._modules.values()) if len(children) == 2 and isinstance(children[1], Sequential) else []
        # (the body's children are run by hand below: forward hooks on the body Sequential itself would not fire)
        hooked = len(children) == 2 and (children[1]._forward_hooks or children[1]._forward_pre_hooks)
        if not body or not isinstance(body[-1], SubmanifoldConvolution) or hooked:
            return ConcatTable.forward(self, input)
        # GradJoin (tensor.py): the two gradients of the block's input -- out of the body's first batch norm, and through
        # the shortcut (identity: the last convolution's incoming gradient itself; NiN: its dInput) -- are added inside
        # one of the two kernels instead of by an autograd add kernel
        from .norm import BatchNormalization
        fits = (input.features.size(0) > 0 and
                (children[0].nOut if isinstance(children[0], NetworkInNetwork) else input.features.size(1)) == body[-1].nOut)
        join = None
        if GRAD_JOIN and fits and isinstance(body[0], BatchNormalization) and self.training and \
                isinstance(children[0], (Identity, NetworkInNetwork)) and not children[0]._forward_hooks:
            join = GradJoin()
        nin_join = join if isinstance(children[0], NetworkInNetwork) else None
        shortcut = children[0](input, join=nin_join) if nin_join is not None else children[0](input)
        x = body[0](input, join=join) if join is not None else body[0](input)
        for mod in body[1:-1]:
            x = mod(x)
        if shortcut.features.size(1) != body[-1].nOut or shortcut.features.size(0) != x.features.size(0):
            assert join is None
            return [shortcut, body[-1](x)]
        return [body[-1](x, residual=shortcut.features, join=join if nin_join is None else None)]


class JoinConcatTable(ConcatTable):
    """ConcatTable(Identity, path) whose results a JoinTable concatenates (the level of scn.UNet,
    sparseconvnet/networkArchitectures.py:294-301): same children, same state_dict keys, same results.  When the path starts
    with a batch norm, the two gradients of the common input -- out of that batch norm, and the left half of the joined
    gradient -- can be added inside the batch norm's apply pass instead of by an autograd add kernel: a GradJoin is handed
    to the batch norm and offered (un-armed) to whoever consumes the table; a JoinTable arms it."""

    @_scoped
    def forward(self, input):
        from .norm import BatchNormalization
        children = list(self._modules.values())
        path = list(children[1]._modules.values()) if len(children) == 2 and isinstance(children[1], Sequential) else []
        hooked = len(children) == 2 and (children[1]._forward_hooks or children[1]._forward_pre_hooks)
        if not (GRAD_JOIN and path and isinstance(children[0], Identity) and isinstance(path[0], BatchNormalization)
                and self.training and not hooked and torch.is_grad_enabled()):
            return ConcatTable.forward(self, input)
        join = GradJoin(armed=False)
        join.left = input.features
        x = path[0](input, join=join)
        for mod in path[1:]:
            x = mod(x)
        x.left_join = join
        return [children[0](input), x]


class _JoinFunction(Function):
    """[left | right] -> joined.  `base` (may be None): a [rows, 2 planes] buffer whose right half `right` already is (a
    Deconvolution with join_right wrote it there): only the left half is copied in.  Backward: the two halves of the
    joined gradient as column slices, no copies; with an armed GradJoin the left one is left for the batch norm at the
    head of the other path to add."""

    @staticmethod
    def forward(ctx, left, right, base, join):
        ctx.p, ctx.join = left.size(1), join
        if base is None:
            return torch.cat([left, right], 1)
        base[:, :left.size(1)].copy_(left)
        ctx.mark_dirty(base)
        return base

    @staticmethod
    def backward(ctx, g):
        g_left, g_right = g[:, :ctx.p], g[:, ctx.p:]
        join, ctx.join = ctx.join, None
        if join is not None and ctx.needs_input_grad[0]:
            other = join.take("join")
            if other is None:
                join.put(g_left, "join")
                g_left = None
            else:
                g_left = g_left + other
        return g_left, g_right, None, None


class JoinTable(torch.nn.Sequential):
    """concatenate feature planes of a list of tensors living on the same sites"""

    def forward(self, input):
        stats = [t.get_colstats() for t in input]  # (before anything is written: they are keyed on the tensors' versions)
        if len(input) == 2 and input[0].features.dim() == 2 and input[0].features.size(0) == input[1].features.size(0) \
                and input[0].features.size(0) > 0:
            left, right = input[0].features, input[1].features
            base = getattr(input[1], "join_base", None)
            if base is not None and not (base.size(1) == left.size(1) + right.size(1) and right.data_ptr() ==
                                         base.data_ptr() + 4 * left.size(1) and left.dtype == base.dtype):
                base = None
            join = getattr(input[1], "left_join", None)
            if join is not None and join.left is not left:
                join = None  # (the table's left entry is not the tensor the batch norm at the head of the path received)
            if join is not None:
                join.armed = True
            out = _like(input[0], apply_function(_JoinFunction, left, right, base, join))
        else:
            out = _like(input[0], torch.cat([t.features for t in input], 1))
        if all(stats):  # the statistics of a concatenation are the concatenation of the statistics
            out.set_colstats([seg for st in stats for seg in st])
        return out

    def input_spatial_size(self, out_size):
        return out_size


class AddTable(torch.nn.Sequential):
    def forward(self, input):
        if len(input) == 1:  # ResidualConcatTable already added the shortcut inside the last convolution
            return input[0]
        total = input[0].features
        for t in input[1:]:
            total = total + t.features
        return _like(input[0], total)

    def input_spatial_size(self, out_size):
        return out_size


class _NiNFunction(Function):
    @staticmethod
    def forward(ctx, features, weight, bias, join=None):
        out = like_empty(features)
        macs = SCN.NetworkInNetwork_updateOutput(features, out, weight, bias)
        counters.add(macs, out.nelement())
        ctx.save_for_backward(features, weight, bias)
        ctx.join = join
        return out

    @staticmethod
    def backward(ctx, grad_output):
        from .conv import grad_owner
        features, weight, bias = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_input = like_empty(grad_output) if ctx.needs_input_grad[0] else None
        # the weight gradient goes straight into the owner's flat buffer when there is one (as the convolutions' does)
        owner = grad_owner(weight)
        direct = owner is not None
        grad_weight = weight.grad if direct else torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        # GradJoin: the other gradient of `features` (out of the body's first batch norm), if that finished first, is added
        # inside the dInput kernel; otherwise this one is left for the batch norm's apply pass to add
        join, ctx.join = ctx.join, None
        extra = join.take("nin") if (join is not None and grad_input is not None) else None
        with SCN.dw_async(direct):
            SCN.NetworkInNetwork_backward(features, grad_input, grad_output, weight, grad_weight, grad_bias,
                                          _d_in_add=extra.contiguous() if extra is not None else None)
        if join is not None and grad_input is not None and extra is None:
            join.put(grad_input, "nin")
            grad_input = None
        if direct:
            owner.grad_written(weight)
            grad_weight = None
        return grad_input, grad_weight, (grad_bias if grad_bias.numel() else None), None


class NetworkInNetwork(Module):
    """1x1 convolution: features @ weight (+ bias); weight [nIn, nOut] ~ N(0, sqrt(2/nIn))"""

    def __init__(self, nIn, nOut, bias):
        Module.__init__(self)
        self.nIn, self.nOut = nIn, nOut
        self.weight = Parameter(torch.empty(nIn, nOut).normal_(0, (2.0 / nIn) ** 0.5))
        if bias:
            self.bias = Parameter(torch.zeros(nOut))

    def forward(self, input, join=None):
        assert input.features.nelement() == 0 or input.features.size(1) == self.nIn, (self.nIn, input.features.shape)
        if SCN._FWD_DEPTH[0] == 0:  # called directly, outside any container's forward (see SCN.py, "Forward scopes")
            SCN.weight_touched(self.weight)
        return _like(input, apply_function(_NiNFunction, input.features, self.weight, optionalTensor(self, "bias"), join))

    def input_spatial_size(self, out_size):
        return out_size

    def __repr__(self):
        return "NetworkInNetwork%d->%d" % (self.nIn, self.nOut)
