"""The whole backbone (InputLayer -> stem -> scn.UNet -> BN -> OutputLayer), forward AND backward, evaluated on the CPU
with the REFERENCE'S OWN compiled loops (oracle/_ref/libscn_ref.so: rule_index_select / at::mm / rule_index_add_ of
sparseconvnet/SCN/CPU/Convolution.cpp:8-33,60-68,104-109, BatchNormalization_ForwardPass/_BackwardPass of
sparseconvnet/SCN/CPU/BatchNormalization.cpp:12-107, InputLayer_*Pass of sparseconvnet/SCN/CPU/IOLayers.cpp:12-47; NetworkInNetwork = mm as
sparseconvnet/SCN/CPU/NetworkInNetwork.cpp:8-46) over the oracle's rulebooks.  TEST INFRASTRUCTURE: bench.py's
``cpu_baseline`` leg ("reference loops") and tests/test_oracle.py (which pins it against oracle/torch_ref.py).

The module tree is walked once; every op returns its output and a closure that maps the gradient of the output to the
gradient of the input (the reference's *_backward functions, called in reverse order, as torch autograd would call
the reference's Functions)."""
import time

import numpy as np

from . import ref as REF
from . import rules as R


class _Ctx(object):
    def __init__(self, coords, mode):
        self.grid0, self.header, self.table, self.site_of_point = R.input_layer_rules(coords, mode=mode)
        self.grids, self.sub, self.strided = {}, {}, {}
        self.mode = mode
        self.masks, self.mask_flips, self.mask_total = None, 0, 0

    def submanifold(self, size):
        if size not in self.sub:
            self.sub[size] = [np.ascontiguousarray(r, np.int32) for r in R.submanifold_rules(self.grids[size])[0]]
        return self.sub[size]

    def conv(self, fine, coarse):
        if fine not in self.strided:
            g, rules, _ = R.strided_rules(self.grids[fine])
            self.grids[coarse] = g
            self.strided[fine] = [np.ascontiguousarray(r, np.int32) for r in rules]
        return self.strided[fine]


class _T(object):
    def __init__(self, f, size):
        self.f, self.size = f, size


def _w(mod, name):
    return np.ascontiguousarray(getattr(mod, name).detach().cpu().numpy(), dtype=REF._DT[0])


def _conv(x, w, lists, n_out, in_col, grads, key, seconds):
    t0 = time.perf_counter()
    out = REF.conv_forward(x, w, lists, n_out, in_col=in_col)
    seconds[0] += time.perf_counter() - t0

    def bwd(g):
        t0 = time.perf_counter()
        d_in, d_w = REF.conv_backward(x, g, w, lists, in_col=in_col)
        seconds[0] += time.perf_counter() - t0
        grads[key] = grads.get(key, 0) + d_w
        return d_in
    return out, bwd


def _run(mod, t, ctx, grads, sec):
    from occuseg_amd.sparseconvnet import containers as C, conv as V, io as IO, norm as NM
    import torch
    if isinstance(mod, V.SubmanifoldConvolution):
        n = ctx.grids[t.size].n
        out, bwd = _conv(t.f, _w(mod, "weight"), ctx.submanifold(t.size), n, 0, grads, (id(mod), "weight"), sec["conv"])
        return _T(out, t.size), bwd
    if isinstance(mod, V.Convolution):
        out_size = tuple((np.array(t.size) - 2) // 2 + 1)
        lists = ctx.conv(t.size, out_size)
        out, bwd = _conv(t.f, _w(mod, "weight"), lists, ctx.grids[out_size].n, 0, grads, (id(mod), "weight"), sec["conv"])
        return _T(out, out_size), bwd
    if isinstance(mod, V.Deconvolution):
        out_size = tuple((np.array(t.size) - 1) * 2 + 2)
        lists = ctx.conv(out_size, t.size)
        out, bwd = _conv(t.f, _w(mod, "weight"), lists, ctx.grids[out_size].n, 1, grads, (id(mod), "weight"), sec["conv"])
        return _T(out, out_size), bwd
    if isinstance(mod, NM.BatchNormalization):
        w = _w(mod, "weight") if getattr(mod, "weight", None) is not None else None
        b = _w(mod, "bias") if getattr(mod, "bias", None) is not None else None
        x = t.f
        t0 = time.perf_counter()
        y, sm, si, _rm, _rv = REF.bn_forward(x, w, b, mod.running_mean.cpu().numpy(), mod.running_var.cpu().numpy(),
                                             mod.eps, mod.momentum, mod.training, mod.leakiness)
        sec["bn"][0] += time.perf_counter() - t0

        # ctx.masks[id(module)] (optional): the activation mask to differentiate with instead of this evaluation's own sign
        # of y -- BatchNormalization_BackwardPass reads y only for its sign (sparseconvnet/SCN/CPU/BatchNormalization.cpp:80-83),
        # so the mask is handed over AS y.  (An element with |y| ~ 1e-7 has a different sign in two f32 evaluations that sum
        # in different orders; with the device's masks both backward passes differentiate the same piecewise-linear function.)
        forced = ctx.masks.get(id(mod)) if getattr(ctx, "masks", None) else None
        if forced is not None:
            ctx.mask_flips += int(((y > 0) != forced).sum())
            ctx.mask_total += int(forced.size)
        y_sign = y if forced is None else np.where(forced, 1.0, -1.0).astype(REF._DT[0])

        def bwd(g):
            t0 = time.perf_counter()
            dx, _d, dw, db = REF.bn_backward(x, y_sign, g, sm, si, w, mod.leakiness)
            sec["bn"][0] += time.perf_counter() - t0
            grads[(id(mod), "weight")], grads[(id(mod), "bias")] = dw, db
            return dx
        return _T(y, t.size), bwd
    if isinstance(mod, C.NetworkInNetwork):
        w = torch.from_numpy(_w(mod, "weight"))
        x = torch.from_numpy(t.f)
        t0 = time.perf_counter()
        out = torch.mm(x, w).numpy()
        sec["nin"][0] += time.perf_counter() - t0

        def bwd(g):
            t0 = time.perf_counter()
            gt = torch.from_numpy(np.ascontiguousarray(g))
            grads[(id(mod), "weight")] = torch.mm(x.t(), gt).numpy()
            d = torch.mm(gt, w.t()).numpy()
            sec["nin"][0] += time.perf_counter() - t0
            return d
        return _T(out, t.size), bwd
    if isinstance(mod, C.Identity):
        return t, lambda g: g
    if isinstance(mod, C.ConcatTable):
        outs, bwds = zip(*[_run(ch, t, ctx, grads, sec) for ch in mod._modules.values()])

        def bwd(gs):
            total = None
            for b, g in zip(bwds, gs):
                d = b(g)
                total = d if total is None else total + d
            return total
        return list(outs), bwd
    if isinstance(mod, C.JoinTable):
        widths = [u.f.shape[1] for u in t]
        out = np.concatenate([u.f for u in t], 1)

        def bwd(g):
            return [np.ascontiguousarray(p) for p in np.split(g, np.cumsum(widths)[:-1], 1)]
        return _T(out, t[0].size), bwd
    if isinstance(mod, C.AddTable):
        total = t[0].f
        for u in t[1:]:
            total = total + u.f
        k = len(t)
        return _T(total, t[0].size), lambda g: [g] * k
    if isinstance(mod, IO.OutputLayer):
        sop = ctx.site_of_point
        n = t.f.shape[0]

        def bwd(g):  # sparseconvnet/SCN/CPU/IOLayers.cpp:87-104: point gradients summed back per site
            d = np.zeros((n, g.shape[1]), REF._DT[0])
            np.add.at(d, sop, g)
            return d
        return t.f[sop], bwd
    if isinstance(mod, torch.nn.Sequential):
        bwds = []
        for ch in mod._modules.values():
            t, b = _run(ch, t, ctx, grads, sec)
            bwds.append(b)

        def bwd(g):
            for b in reversed(bwds):
                g = b(g)
            return g
        return t, bwd
    raise NotImplementedError(type(mod))


def forward_backward(net, coords, feats, threads=None, ctx=None, masks=None, dtype=np.float32):
    """one forward + backward (loss = sum of the outputs) of ``net`` (Sequential starting with an InputLayer).
    masks: {id(batch-norm module): bool array [N, C]} = activation masks to use in the backward pass (see _run).
    dtype: np.float32 = the arithmetic the reference runs; np.float64 = the same templates of the reference instantiated at
    double (oracle/ref_harness.cpp: *_f64) -- the reference's loops without their float32 rounding.
    Returns (output [P, m], {(id(module), name): gradient}, seconds by group, ctx): ``seconds['total']`` is the wall
    time of the arithmetic (rulebook generation excluded: ctx can be passed in pre-built, and lists are generated
    before the clock starts)."""
    from occuseg_amd.sparseconvnet import conv as V, io as IO
    if np.dtype(dtype) != np.dtype(REF._DT[0]):
        with REF.precision(dtype):
            return forward_backward(net, coords, feats, threads, ctx, masks, dtype)
    if threads is not None:
        REF._load().ref_set_threads(int(threads))
        import torch
        torch.set_num_threads(int(threads))
    children = list(net._modules.values())
    inp = children[0]
    assert isinstance(inp, IO.InputLayer)
    size = tuple(int(v) for v in inp.spatial_size.tolist())
    if ctx is None:
        ctx = _Ctx(np.asarray(coords), inp.mode)
        ctx.grids[size] = ctx.grid0
        s = size  # every rulebook the network will ask for, before the clock starts
        for mod in net.modules():
            if isinstance(mod, V.Convolution) and s not in ctx.strided:
                ctx.submanifold(s)
                nxt = tuple((np.array(s) - 2) // 2 + 1)
                ctx.conv(s, nxt)
                s = nxt
        ctx.submanifold(s)
    sec = {"conv": [0.0], "bn": [0.0], "nin": [0.0]}
    grads = {}
    ctx.masks, ctx.mask_flips, ctx.mask_total = masks, 0, 0
    t0 = time.perf_counter()
    x = REF.input_forward(feats, ctx.table, mode=inp.mode)
    t = _T(x, size)
    bwds = []
    for ch in children[1:]:
        t, b = _run(ch, t, ctx, grads, sec)
        bwds.append(b)
    out = t.f if isinstance(t, _T) else t
    g = np.ones_like(out)
    for b in reversed(bwds):
        g = b(g)
    REF.input_backward(g, ctx.table, len(feats), mode=inp.mode)
    total = time.perf_counter() - t0
    seconds = {k: v[0] for k, v in sec.items()}
    seconds["total"] = total
    return out, grads, seconds, ctx
