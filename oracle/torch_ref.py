"""Oracle: whole-network CPU evaluation with torch autograd.  TEST INFRASTRUCTURE.

Walks a module tree built from ``occuseg_amd.sparseconvnet`` (weights are read
from it, nothing of its compute path is used) and evaluates it on the CPU with
the oracle rulebooks (``oracle.rules``) and the reference's own formulation of
the arithmetic -- index_select -> mm -> index_add_ per kernel offset
(sparseconvnet/SCN/CPU/Convolution.cpp:60-68, the commented lines are exactly these calls) and the
batch-norm formulas of sparseconvnet/SCN/CPU/BatchNormalization.cpp:12-62.  Backward comes from
torch autograd, i.e. it is independent of the hand-derived backward formulas in
``oracle.arith`` and of the HIP kernels.

Used (a) in float64 as the end-to-end parity reference of tests, (b) in float32
with all host cores as ``bench.py``'s ``cpu_baseline`` ("port": same
gather/sgemm/scatter work as the reference CPU path, at::mm == torch.mm).
"""
import time

import numpy as np
import torch

from . import rules as R


class _Ctx(object):
    """oracle Metadata: grids and rulebooks per spatial size"""

    def __init__(self, coords, mode):
        t0 = time.perf_counter()
        self.grid0, self.header, self.table, self.site_of_point = R.input_layer_rules(coords, mode=mode)
        self.grids = {}
        self.sub = {}
        self.strided = {}
        self.n_points = len(coords)
        self.mode = mode
        self.rule_seconds = time.perf_counter() - t0  # time spent generating rulebooks (numpy)

    def submanifold(self, size):
        if size not in self.sub:
            t0 = time.perf_counter()
            rules, _ = R.submanifold_rules(self.grids[size])
            self.sub[size] = [(torch.from_numpy(r[:, 0].astype(np.int64)), torch.from_numpy(r[:, 1].astype(np.int64)))
                              for r in rules]
            self.rule_seconds += time.perf_counter() - t0
        return self.sub[size]

    def conv(self, fine_size, coarse_size):
        if fine_size not in self.strided:
            t0 = time.perf_counter()
            coarse, rules, _ = R.strided_rules(self.grids[fine_size])
            self.grids[coarse_size] = coarse
            self.strided[fine_size] = [(torch.from_numpy(r[:, 0].astype(np.int64)),
                                        torch.from_numpy(r[:, 1].astype(np.int64))) for r in rules]
            self.rule_seconds += time.perf_counter() - t0
        return self.strided[fine_size]


class _T(object):
    def __init__(self, f, size):
        self.f, self.size = f, size


def _rule_conv(x, w, lists, n_out, in_col, out_col):
    out = torch.zeros((n_out, w.shape[2]), dtype=x.dtype)
    for k, cols in enumerate(lists):
        if cols[0].numel():
            out = out.index_add(0, cols[out_col], x.index_select(0, cols[in_col]) @ w[k])
    return out


def _bn(mod, x, dtype, stats):
    n = x.shape[0]
    g = _param(mod, "weight", dtype, stats) if getattr(mod, "weight", None) is not None else None
    b = _param(mod, "bias", dtype, stats) if getattr(mod, "bias", None) is not None else None
    if mod.training:
        mean = x.sum(0) / n
        s = (x * x).sum(0) - mean * mean * n
        invstd = (s / n + mod.eps) ** -0.5
        stats["bn"][id(mod)] = (mean.detach(), (s / (n - 1)).detach() if n > 1 else None)
    else:
        mean = mod.running_mean.cpu().to(dtype)
        invstd = (mod.running_var.cpu().to(dtype) + mod.eps) ** -0.5
    w = invstd * (g if g is not None else 1.0)
    y = x * w + ((b if b is not None else 0.0) - mean * w)
    # stats["masks"][id(module)]: activation mask to USE instead of this evaluation's own sign of y (the device's mask:
    # an element with |y| ~ 1e-7 has a different sign in fp32 and fp64, and one flipped mask moves every gradient
    # upstream of it by a whole term -- with the device's masks the backward pass can be compared at 1e-4)
    pos = stats["masks"].get(id(mod)) if stats.get("masks") else None
    stats.setdefault("own_masks", {})[id(mod)] = (y > 0).detach()
    return torch.where(y > 0 if pos is None else pos, y, y * mod.leakiness)


def _param(mod, name, dtype, stats):
    d = stats["params"].setdefault(id(mod), {})
    if name not in d:
        d[name] = getattr(mod, name).detach().cpu().to(dtype).clone().requires_grad_(True)
    return d[name]


def _run(mod, t, ctx, dtype, stats):
    from occuseg_amd.sparseconvnet import containers as C, conv as V, io as IO, norm as NM
    if isinstance(mod, IO.InputLayer):
        raise RuntimeError("InputLayer is handled by ref_forward")
    if isinstance(mod, V.SubmanifoldConvolution):
        w = _param(mod, "weight", dtype, stats)
        n = ctx.grids[t.size].n
        return _T(_rule_conv(t.f, w, ctx.submanifold(t.size), n, 0, 1), t.size)
    if isinstance(mod, V.Convolution):
        w = _param(mod, "weight", dtype, stats)
        out_size = tuple((np.array(t.size) - 2) // 2 + 1)
        lists = ctx.conv(t.size, out_size)
        return _T(_rule_conv(t.f, w, lists, ctx.grids[out_size].n, 0, 1), out_size)
    if isinstance(mod, V.Deconvolution):
        w = _param(mod, "weight", dtype, stats)
        out_size = tuple((np.array(t.size) - 1) * 2 + 2)
        lists = ctx.conv(out_size, t.size)
        return _T(_rule_conv(t.f, w, lists, ctx.grids[out_size].n, 1, 0), out_size)
    if isinstance(mod, NM.BatchNormalization):
        return _T(_bn(mod, t.f, dtype, stats), t.size)
    if isinstance(mod, C.NetworkInNetwork):
        return _T(t.f @ _param(mod, "weight", dtype, stats), t.size)
    if isinstance(mod, C.Identity):
        return t
    if isinstance(mod, C.ConcatTable):
        return [_run(ch, t, ctx, dtype, stats) for ch in mod._modules.values()]
    if isinstance(mod, C.JoinTable):
        return _T(torch.cat([u.f for u in t], 1), t[0].size)
    if isinstance(mod, C.AddTable):
        total = t[0].f
        for u in t[1:]:
            total = total + u.f
        return _T(total, t[0].size)
    if isinstance(mod, IO.OutputLayer):
        return t.f.index_select(0, torch.from_numpy(ctx.site_of_point))
    if isinstance(mod, torch.nn.Sequential):
        for ch in mod._modules.values():
            t = _run(ch, t, ctx, dtype, stats)
        return t
    raise NotImplementedError(type(mod))


def ref_forward(net, coords, feats, dtype=torch.float64, masks=None):
    """net: Sequential whose first child is an InputLayer.  Returns (output tensor with grad graph,
    stats) where stats['params'][id(module)][name] are leaf tensors holding the grads after backward,
    stats['bn'][id(module)] = (batch mean, unbiased batch var), stats['own_masks'][id(module)] = this evaluation's
    y > 0.  masks: {id(batch-norm module): bool tensor} = activation masks to use instead (see _bn)."""
    from occuseg_amd.sparseconvnet import io as IO
    children = list(net._modules.values())
    inp = children[0]
    assert isinstance(inp, IO.InputLayer)
    ctx = _Ctx(np.asarray(coords), inp.mode)
    size = tuple(int(v) for v in inp.spatial_size.tolist())
    ctx.grids[size] = ctx.grid0
    f = torch.as_tensor(np.asarray(feats)).to(dtype)
    # out[s] = sum_j mult * in[p_j]   (sparseconvnet/SCN/CPU/IOLayers.cpp:12-30)
    table = ctx.table
    cnt = torch.from_numpy(table[:, 0].astype(np.int64))
    x = torch.zeros((table.shape[0], f.shape[1]), dtype=dtype)
    mult = torch.where(cnt > 0, 1.0 / cnt.clamp(min=1).to(dtype), torch.ones_like(cnt, dtype=dtype)) \
        if inp.mode == 4 else torch.ones(table.shape[0], dtype=dtype)
    for j in range(table.shape[1] - 1):
        m = cnt > j
        idx = torch.from_numpy(table[:, 1 + j].astype(np.int64))[m]
        x[m] += mult[m, None] * f[idx]
    stats = {"params": {}, "bn": {}, "ctx": ctx, "masks": masks}
    t = _T(x, size)
    for ch in children[1:]:
        t = _run(ch, t, ctx, dtype, stats)
    return (t.f if isinstance(t, _T) else t), stats
