"""BatchNormalization over active sites with the fused (leaky) ReLU.

Operator API of sparseconvnet/batchNormalization.py:13-161: eps 1e-4, momentum 0.9 with the
reference's convention running = momentum*running + (1-momentum)*batch,
`leakiness` 0 = ReLU, 1 = no activation; buffers `running_mean`/`running_var`,
parameters `weight`/`bias`.
"""
import torch
from torch.autograd import Function
from torch.nn import Module, Parameter

from .. import SCN
from . import conv
from .tensor import SparseConvNetTensor, apply_function, like_empty, optionalTensor, optionalTensorReturn


class _BatchNormFunction(Function):
    @staticmethod
    def forward(ctx, features, weight, bias, running_mean, running_var, eps, momentum, train, leakiness, colstats=None,
                link_out=None, join=None):
        ctx.train, ctx.leakiness = train, leakiness
        ctx.join = join
        out = like_empty(features)
        save_mean = like_empty(features)
        save_invstd = like_empty(features)
        SCN.BatchNormalization_updateOutput(features, out, save_mean, save_invstd, running_mean, running_var,
                                            weight, bias, eps, momentum, train, leakiness, _colstats=colstats)
        ctx.save_for_backward(features, out, weight, bias, running_mean, running_var, save_mean, save_invstd)
        ctx.link = None
        if link_out is not None and train:  # for the convolution that consumes `out` (SCN.BnLink)
            ctx.link = SCN.BnLink(features, save_mean, save_invstd, weight, bias, leakiness)
            link_out.append(ctx.link)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, out, weight, bias, running_mean, running_var, save_mean, save_invstd = ctx.saved_tensors
        assert ctx.train
        # the reference backend rewrites d_out in place (sparseconvnet/SCN/CUDA/BatchNormalization.cu:151-153), and
        # autograd's buffer must never be mutated: ask the backend to recompute the activation mask from x (the
        # forward's own fma: bit-identical sign) -- d_out is only read, y is not read at all (20 B/element, not 28)
        grad_output = grad_output.contiguous()
        grad_input = like_empty(grad_output)
        # fused parameter-gradient accumulation, as in conv.py (the owner of the .grad buffers opts in)
        owner = conv.grad_owner(weight, bias) if weight.numel() > 0 and bias.numel() > 0 else None
        direct = owner is not None
        grad_weight = weight.grad if direct else torch.zeros_like(weight)
        grad_bias = bias.grad if direct else torch.zeros_like(bias)
        # (not in the reference) the two reductions of this pass, if the convolution that consumed `out` took them from
        # its dInput tile (SCN.BnLink), and the gradient arriving at `features` along the other path of a residual block
        # (GradJoin): added inside the apply pass instead of by an autograd add kernel
        partials = ctx.link.take(grad_output) if ctx.link is not None else None
        join, extra = ctx.join, None
        if join is not None and not join.armed:
            join = None  # nobody took the other end (no JoinTable behind the JoinConcatTable): plain path
        if join is not None and join.pending is not None:
            extra = join.take("bn")
            if extra is not None and not (extra.shape == features.shape and (extra.is_contiguous() or SCN._row_slice(extra))):
                extra = extra.contiguous()
        SCN.BatchNormalization_backward(features, grad_input, out, grad_output, save_mean, save_invstd,
                                        running_mean, running_var, weight, bias, grad_weight, grad_bias,
                                        ctx.leakiness, _rewrite_d_output=False, _accumulate=direct, _mask_from_x=True,
                                        _partials=partials, _d_x_add=extra)
        ctx.link = ctx.join = None
        if join is not None and extra is None:  # first of the two paths to finish: the other one adds this
            join.put(grad_input, "bn")
            grad_input = None
        if direct:
            owner.grad_written(weight)
            owner.grad_written(bias)
            return (grad_input, None, None) + (None,) * 9
        return (grad_input, optionalTensorReturn(grad_weight), optionalTensorReturn(grad_bias)) + (None,) * 9


class BatchNormalization(Module):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1):
        Module.__init__(self)
        self.nPlanes, self.eps, self.momentum = nPlanes, eps, momentum
        self.affine, self.leakiness = affine, leakiness
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        if affine:
            self.weight = Parameter(torch.ones(nPlanes))
            self.bias = Parameter(torch.zeros(nPlanes))

    def forward(self, input, join=None):
        """``join`` (not in the reference; used by the residual blocks of scn.UNet): see tensor.GradJoin"""
        assert input.features.nelement() == 0 or input.features.size(1) == self.nPlanes, \
            (self.nPlanes, input.features.shape)
        output = SparseConvNetTensor(metadata=input.metadata, spatial_size=input.spatial_size)
        # statistics the producing convolution(s) emitted for exactly this tensor (training only): no second pass over it
        ext = input.get_colstats() if self.training and hasattr(input, "get_colstats") else None
        if ext and sum(n for n, _ in ext) != self.nPlanes:
            ext = None
        link_out = [] if (self.training and torch.is_grad_enabled() and input.features.requires_grad) else None
        output.features = apply_function(_BatchNormFunction, input.features, optionalTensor(self, "weight"), optionalTensor(self, "bias"), self.running_mean,
            self.running_var, self.eps, self.momentum, self.training, self.leakiness, ext, link_out, join)
        if link_out:
            output.set_bn_link(link_out[0])
        return output

    def input_spatial_size(self, out_size):
        return out_size

    def _name(self):
        return "BatchNorm"

    def __repr__(self):
        s = "%s(%d,eps=%s,momentum=%s,affine=%s" % (self._name(), self.nPlanes, self.eps, self.momentum, self.affine)
        if self.leakiness > 0 and type(self) is not BatchNormReLU:
            s += ",leakiness=%s" % self.leakiness
        return s + ")"


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0)

    def _name(self):
        return "BatchNormReLU"


class BatchNormLeakyReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, leakiness)

    def _name(self):
        return "BatchNormLeakyReLU"
