"""SubmanifoldConvolution / Convolution / Deconvolution modules.

Operator API of sparseconvnet/submanifoldConvolution.py:18-128, sparseconvnet/convolution.py:14-127,
sparseconvnet/deconvolution.py:13-155: same constructor arguments, parameter names/shapes
(`weight` [filter_volume, nIn, nOut] ~ N(0, sqrt(2/(nIn*volume))), optional `bias`),
same forward contract on SparseConvNetTensor.  One autograd Function serves
the three ops; it calls the SCN boundary functions with the reference's names.
"""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from .. import SCN
from . import counters
from .tensor import SparseConvNetTensor, apply_function, like_empty, optionalTensor, optionalTensorReturn, toLongTensor


def grad_owner(*params):
    """Opt-in "fused parameter-gradient accumulation" (see _RuleConvFunction.backward): the owner of the (zeroed) .grad
    buffers (occuseg_amd.dp.FlatGradBucket) marks ITS parameters with ``_scn_grad_owner``; the backward kernels then
    accumulate straight into ``param.grad`` and report to the owner.  Per parameter, nothing process-global; only
    valid when gradients are consumed through .grad (loss.backward()), not through torch.autograd.grad.
    Returns the owner when every given parameter has the same one and its .grad is still the owner's view."""
    owner = getattr(params[0], "_scn_grad_owner", None)
    if owner is None:
        return None
    for p in params:
        if getattr(p, "_scn_grad_owner", None) is not owner or not owner.owns(p):
            return None
    return owner


JOIN_IN_PLACE = True  # A/B switch: a Deconvolution marked `join_right` (scn.UNet's) writes into the right half of the joined tensor
EMIT_COLSTATS = True  # A/B switch: convolutions hand their output's column statistics to the following batch norm


class _RuleConvFunction(Function):
    """kind: 'submanifold' | 'conv' | 'deconv'"""

    @staticmethod
    def forward(ctx, features, weight, bias, kind, metadata, in_size, out_size, filter_size, filter_stride,
                dilated_rate, residual=None, stats_out=None, bn_link=None, join=None, join_base=None):
        ctx.kind, ctx.scn_meta, ctx.dilated_rate = kind, metadata, dilated_rate
        ctx.bn_link, ctx.join = bn_link, join
        ctx.sizes = (in_size, out_size, filter_size, filter_stride)
        out = like_empty(features)
        if kind == "submanifold":
            # residual: out = conv(features) + residual in one kernel (the shortcut of a residual block)
            macs = SCN.SubmanifoldConvolution_updateOutput(in_size, filter_size, metadata, features, out, weight,
                                                           bias, dilated_rate, residual=residual, stats_out=stats_out)
        elif kind == "conv":
            macs = SCN.Convolution_updateOutput(in_size, out_size, filter_size, filter_stride, metadata, features,
                                                out, weight, bias, stats_out=stats_out)
        else:
            # join_base (a list that receives it): the result is written straight into the RIGHT half of a [rows, 2 nOut]
            # buffer -- the JoinTable that follows copies only the other half in (`_JoinFunction`) -- when the launch is one
            # that can write a column slice
            view = None
            if join_base is not None and features.size(0) > 0 and SCN.deconvolution_can_write_a_slice(
                    in_size, out_size, filter_size, filter_stride, metadata, weight.size(1), weight.size(2)):
                n_out, co = metadata.getNActive(out_size), weight.size(2)
                base = torch.empty((n_out, 2 * co), dtype=features.dtype, device=features.device)
                view = base[:, co:]
                join_base.append(base)
            macs = SCN.Deconvolution_updateOutput(in_size, out_size, filter_size, filter_stride, metadata, features,
                                                  out, weight, bias, stats_out=stats_out, _out_view=view)
            if view is not None:
                out = view
        counters.add(macs, out.nelement())
        ctx.save_for_backward(features, weight, bias)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, weight, bias = ctx.saved_tensors
        in_size, out_size, filter_size, filter_stride = ctx.sizes
        if not (SCN._row_slice(grad_output) and not grad_output.is_contiguous()):  # (a column slice of a joined gradient is read in place)
            grad_output = grad_output.contiguous()
        # the reference always computes dInput; it is skipped here when autograd does not ask for it
        grad_input = torch.empty(0, dtype=grad_output.dtype, device=grad_output.device) if ctx.needs_input_grad[0] else None
        # The dWeight kernel ACCUMULATES (float atomics into a zeroed buffer, as the reference's does).  When the
        # owner of the gradients says so (occuseg_amd.dp.FlatGradBucket: .grad are views of one zeroed flat buffer,
        # see grad_owner above) it accumulates straight into weight.grad: no zeros_like, no autograd add kernel.
        owner = grad_owner(weight)
        direct = owner is not None
        grad_weight = weight.grad if direct else torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        m = ctx.scn_meta
        # direct accumulation: nothing reads the weight gradient before the end of the backward pass, so the dWeight
        # kernel may run on a side stream beside what follows (SCN.dw_async keeps grad_output referenced until that
        # kernel has finished, so autograd cannot add another gradient into it in place meanwhile -- the shortcut of a
        # residual block, add_feature_planes, AddTable: whatever shares this tensor)
        # bn_link: `features` is the output of a batch norm; the dInput launch below leaves that batch norm's two backward
        # reductions with it (SCN.BnLink)
        link = ctx.bn_link if grad_input is not None else None
        with SCN.dw_async(direct):
            if ctx.kind == "submanifold":
                SCN.SubmanifoldConvolution_backward(in_size, filter_size, m, features, grad_input, grad_output, weight,
                                                    grad_weight, grad_bias, ctx.dilated_rate, _bn_link=link)
            elif ctx.kind == "conv":
                SCN.Convolution_backward(in_size, out_size, filter_size, filter_stride, m, features, grad_input,
                                         grad_output, weight, grad_weight, grad_bias, _bn_link=link)
            else:
                SCN.Deconvolution_backward(in_size, out_size, filter_size, filter_stride, m, features, grad_input,
                                           grad_output, weight, grad_weight, grad_bias, _bn_link=link)
        ctx.scn_meta = ctx.bn_link = None
        if direct:
            owner.grad_written(weight)
            grad_weight = None
        # d(out)/d(residual) = identity: the incoming gradient itself.  With a GradJoin (the residual IS the block's
        # input x, whose other gradient comes out of the body's first batch norm) it is left for that batch norm's
        # apply pass to add; if that one finished first, the sum is formed here.
        grad_residual = grad_output if ctx.needs_input_grad[10] else None
        join, ctx.join = ctx.join, None
        if join is not None and grad_residual is not None:
            other = join.take("conv")
            if other is None:
                join.put(grad_residual, "conv")
                grad_residual = None
            else:
                grad_residual = grad_residual + other
        return (grad_input, grad_weight, optionalTensorReturn(grad_bias)) + (None,) * 7 + (grad_residual, None, None, None, None)


class _RuleConv(Module):
    kind = None
    label = None

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias, dilated_rate=1):
        Module.__init__(self)
        self.dimension = dimension
        self.nIn, self.nOut = nIn, nOut
        self.filter_size = toLongTensor(dimension, filter_size)
        self.filter_volume = int(self.filter_size.prod().item())
        self.filter_stride = None if filter_stride is None else toLongTensor(dimension, filter_stride)
        self.dilated_rate = dilated_rate
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = Parameter(torch.empty(self.filter_volume, nIn, nOut).normal_(0, std))
        if bias:
            self.bias = Parameter(torch.zeros(nOut))

    def output_spatial_size(self, in_size):
        raise NotImplementedError

    # weights that change without an optimizer step or a backward pass in between: a new weight epoch (round-3 advisor)
    def _load_from_state_dict(self, *args, **kwargs):
        Module._load_from_state_dict(self, *args, **kwargs)
        SCN.weights_changed()

    def _apply(self, fn, *args, **kwargs):
        res = Module._apply(self, fn, *args, **kwargs)
        SCN.weights_changed()
        return res

    def forward(self, input, residual=None, join=None):
        """``residual`` (submanifold only; used by the residual blocks of scn.UNet): features added to the output
        inside the convolution kernel; ``join``: see tensor.GradJoin (the residual is the block's input)"""
        assert input.features.nelement() == 0 or input.features.size(1) == self.nIn, (self.nIn, self.nOut, input)
        assert residual is None or self.kind == "submanifold"
        if SCN._FWD_DEPTH[0] == 0:  # called directly, outside any container's forward: its weights may have been written since
            SCN.weight_touched(self.weight)
        out_size = self.output_spatial_size(input.spatial_size)
        output = SparseConvNetTensor(metadata=input.metadata, spatial_size=out_size)
        # training: the kernel also emits the column statistics of its result (taken from the output tile while it is in
        # LDS) for the batch norm that usually follows
        sink = [] if (self.training and EMIT_COLSTATS) else None
        base = [] if (self.kind == "deconv" and getattr(self, "join_right", False) and JOIN_IN_PLACE) else None
        output.features = apply_function(_RuleConvFunction, input.features, self.weight, optionalTensor(self, "bias"), self.kind, input.metadata,
            input.spatial_size, out_size, self.filter_size, self.filter_stride, self.dilated_rate, residual, sink,
            input.get_bn_link() if hasattr(input, "get_bn_link") else None, join if residual is not None else None, base)
        if sink:
            output.set_colstats([(self.nOut, sink[0])])
        if base:
            output.join_base = base[0]  # features IS its right half (see containers.JoinTable)
        return output

    def _fmt(self, t):
        v = t.tolist()
        return str(v[0]) if min(v) == max(v) else "(" + ",".join(str(i) for i in v) + ")"

    def __repr__(self):
        s = "%s %d->%d C%s" % (self.label, self.nIn, self.nOut, self._fmt(self.filter_size))
        if self.filter_stride is not None:
            s += "/" + self._fmt(self.filter_stride)
        return s


class SubmanifoldConvolution(_RuleConv):
    """scn.SubmanifoldConvolution(dimension, nIn, nOut, filter_size, bias, dilated_rate=1)"""
    kind, label = "submanifold", "SubmanifoldConvolution"

    def __init__(self, dimension, nIn, nOut, filter_size, bias, dilated_rate=1):
        _RuleConv.__init__(self, dimension, nIn, nOut, filter_size, None, bias, dilated_rate)

    def output_spatial_size(self, in_size):
        return in_size

    def input_spatial_size(self, out_size):
        return out_size


class ValidConvolution(SubmanifoldConvolution):
    pass


class Convolution(_RuleConv):
    """scn.Convolution(dimension, nIn, nOut, filter_size, filter_stride, bias)"""
    kind, label = "conv", "Convolution"

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        _RuleConv.__init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias)

    def output_spatial_size(self, in_size):
        # integer division: sparseconvnet/convolution.py:35-38 relied on LongTensor `/` flooring (torch 1.1)
        out = (in_size - self.filter_size) // self.filter_stride + 1
        # same check, same place as the reference's forward (sparseconvnet/convolution.py:36-37): a spatial size the
        # stride does not divide is an AssertionError there too
        assert ((out - 1) * self.filter_stride + self.filter_size == in_size).all(), \
            (in_size, out, self.filter_size, self.filter_stride)
        return out

    def input_spatial_size(self, out_size):
        return (out_size - 1) * self.filter_stride + self.filter_size


class Deconvolution(_RuleConv):
    """scn.Deconvolution(dimension, nIn, nOut, filter_size, filter_stride, bias)"""
    kind, label = "deconv", "Deconvolution"

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        _RuleConv.__init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias)

    def output_spatial_size(self, in_size):
        return (in_size - 1) * self.filter_stride + self.filter_size

    def input_spatial_size(self, out_size):
        return (out_size - self.filter_size) // self.filter_stride + 1
