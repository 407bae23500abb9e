"""Segment reductions over instance labels -- the loss-side hot spots after the backbone.

The reference's losses call torch_scatter 1.4.0 (p1.yml:167) with dim = 0 on [points, planes] tensors:
``scatter_mean`` / ``scatter_std`` per instance in examples/ScanNet/train_instance.py:224-229 and
examples/ScanNet/discriminative.py:52-53,146, ``scatter_add`` / ``scatter_max`` in examples/ScanNet/evaluate_instance.py.
Same names, argument meaning and results here (dim 0 only, float32, GPU only), differentiable, computed by the
segment kernels of libscn_hip.so (LDS partial tables + one global atomic per touched cell) instead of one atomic per
element.  ``index`` may be [P] or broadcast-shaped [P, C] as torch_scatter allows (all columns then equal).
"""
import torch
from torch.autograd import Function

from ._lib import check, load, ptr, stream


def _prep(src, index, dim, dim_size):
    if dim not in (0, -src.dim()):
        raise NotImplementedError("occuseg_amd.segment: only dim = 0 (rows = points), as the reference's losses use it")
    if not src.is_cuda or src.dtype != torch.float32:
        raise RuntimeError("occuseg_amd.segment: float32 GPU tensors only -- this backend has no CPU path")
    shape = src.shape
    s2 = src.reshape(shape[0], -1).contiguous()
    if index.dim() > 1:
        index = index.reshape(index.shape[0], -1)[:, 0]
    index = index.to(torch.int64).contiguous()
    assert index.numel() == shape[0], (index.shape, shape)
    S = int(dim_size) if dim_size is not None else (int(index.max().item()) + 1 if index.numel() else 0)
    return s2, index, S, shape


def _reduce(s2, index, S, op=0):
    P, C = s2.shape
    out = torch.full((S, C), 0.0 if op == 0 else -3.402823466e+38, dtype=torch.float32, device=s2.device)
    check(load().scn_segment_reduce(ptr(s2), ptr(index), P, C, S, op, ptr(out), stream()))
    return out


def _count(index, S, device):
    cnt = torch.zeros(S, dtype=torch.float32, device=device)
    check(load().scn_segment_count(ptr(index), index.numel(), S, ptr(cnt), stream()))
    return cnt


def _gather(table, index, scale=None):
    S, C = table.shape
    out = torch.empty((index.numel(), C), dtype=torch.float32, device=table.device)
    check(load().scn_segment_gather(ptr(table.contiguous()), ptr(index), ptr(scale), index.numel(), C, S, ptr(out),
                                    stream()))
    return out


class _ScatterAdd(Function):
    @staticmethod
    def forward(ctx, s2, index, S):
        ctx.save_for_backward(index)
        return _reduce(s2, index, S, 0)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return _gather(g.contiguous(), index), None, None


class _ScatterMean(Function):
    @staticmethod
    def forward(ctx, s2, index, S):
        cnt = _count(index, S, s2.device).clamp_(min=1)
        ctx.save_for_backward(index, cnt)
        return _reduce(s2, index, S, 0) / cnt.view(-1, 1)

    @staticmethod
    def backward(ctx, g):
        index, cnt = ctx.saved_tensors
        return _gather(g.contiguous(), index, (1.0 / cnt).contiguous()), None, None


class _ScatterStd(Function):
    """torch_scatter 1.4.0 scatter_std: sqrt( sum (x - mean)^2 / clamp(count - unbiased, 1) )"""

    @staticmethod
    def forward(ctx, s2, index, S, unbiased):
        P, C = s2.shape
        cnt = _count(index, S, s2.device)
        mean = _reduce(s2, index, S, 0) / cnt.clamp(min=1).view(-1, 1)
        sq = torch.zeros((S, C), dtype=torch.float32, device=s2.device)
        check(load().scn_segment_sqdev(ptr(s2), ptr(mean), ptr(index), P, C, S, ptr(sq), stream()))
        denom = (cnt - 1 if unbiased else cnt).clamp(min=1)
        std = torch.sqrt(sq / denom.view(-1, 1))
        ctx.save_for_backward(s2, index, mean, std, denom)
        return std

    @staticmethod
    def backward(ctx, g):
        s2, index, mean, std, denom = ctx.saved_tensors
        # d std / d x_p = (x_p - mean) / (denom * std)   (the mean's own dependence on x_p sums to zero)
        coef = g / (std * denom.view(-1, 1))
        coef = torch.where(std > 0, coef, torch.zeros_like(coef))
        return (s2 - _gather(mean, index)) * _gather(coef.contiguous(), index), None, None, None


def scatter_add(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    assert out is None and fill_value == 0
    s2, index, S, shape = _prep(src, index, dim, dim_size)
    return _ScatterAdd.apply(s2, index, S).reshape((S,) + tuple(shape[1:]))


def scatter_mean(src, index, dim=-1, out=None, dim_size=None, fill_value=0):
    assert out is None and fill_value == 0
    s2, index, S, shape = _prep(src, index, dim, dim_size)
    return _ScatterMean.apply(s2, index, S).reshape((S,) + tuple(shape[1:]))


def scatter_std(src, index, dim=-1, out=None, dim_size=None, unbiased=True):
    assert out is None
    s2, index, S, shape = _prep(src, index, dim, dim_size)
    return _ScatterStd.apply(s2, index, S, bool(unbiased)).reshape((S,) + tuple(shape[1:]))


def scatter_max(src, index, dim=-1, out=None, dim_size=None, fill_value=None):
    """returns (max, argmax); segments without a row hold fill_value (torch_scatter: 0 when None) and argmax = P.
    Not differentiable here (the reference uses it in evaluation only, examples/ScanNet/evaluate_instance.py:252,374)."""
    assert out is None
    s2, index, S, shape = _prep(src.detach(), index, dim, dim_size)
    mx = _reduce(s2, index, S, 1)
    P, C = s2.shape
    hit = _gather(mx, index) == s2
    rows = torch.arange(P, device=s2.device).view(-1, 1).expand(P, C)
    arg = torch.full((S, C), P, dtype=torch.int64, device=s2.device)
    arg.scatter_reduce_(0, index.view(-1, 1).expand(P, C), torch.where(hit, rows, torch.full_like(rows, P)), "amin")
    empty = _count(index, S, s2.device) == 0
    mx[empty] = 0.0 if fill_value is None else float(fill_value)
    return mx.reshape((S,) + tuple(shape[1:])), arg.reshape((S,) + tuple(shape[1:]))
