"""CPU: tensor.GradJoin -- the two input gradients of a residual block met inside a kernel instead of by an autograd add -- under
the backward passes that are not the plain one: nested (re-entrant) passes, failed passes repeated over a retained graph, a side
that comes twice (round-4 advisor: a gradient left by another pass was dropped silently)."""
import pytest
import torch

from occuseg_amd.sparseconvnet.tensor import GradJoin


class _Side(torch.autograd.Function):
    """y = scale * x whose input gradient goes through a GradJoin exactly as the modules' backward functions do"""

    @staticmethod
    def forward(ctx, x, join, side, scale, boom):
        ctx.join, ctx.side, ctx.scale, ctx.boom = join, side, scale, boom
        return x * scale

    @staticmethod
    def backward(ctx, g):
        if ctx.boom and ctx.boom[0]:
            raise RuntimeError("boom")
        gx = g * ctx.scale
        other = ctx.join.take(ctx.side)
        if other is None:
            ctx.join.put(gx, ctx.side)
            gx = None
        else:
            gx = gx + other
        return gx, None, None, None, None


class _Nested(torch.autograd.Function):
    """runs the backward of an inner graph in a pass of its own (what re-entrant checkpointing does)"""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        ctx.save_for_backward(x)
        return fn(x.detach()).detach()

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        with torch.enable_grad():
            xi = x.detach().requires_grad_(True)
            out = ctx.fn(xi)
            torch.autograd.backward(out, g)
        return xi.grad, None


def test_both_sides_in_one_pass_either_order():
    for order in (0, 1):
        x = torch.ones(3, requires_grad=True)
        j = GradJoin()
        a = _Side.apply(x, j, "bn", 2.0, None)
        b = _Side.apply(x, j, "conv", 5.0, None)
        ((a * (1 + order)).sum() + (b * (2 - order)).sum()).backward()
        want = 2.0 * (1 + order) + 5.0 * (2 - order)
        assert torch.equal(x.grad, torch.full((3,), want)) and j.pending is None


def test_first_side_in_a_nested_pass_is_added_not_dropped():
    """the side inside the nested pass finishes (and its pass completes) before the outer side runs: round 4 dropped it"""
    x = torch.ones(3, requires_grad=True)
    j = GradJoin()
    # (the engine runs the node that was created LAST first: the nested one)
    outer = _Side.apply(x, j, "conv", 5.0, None)
    inner = _Nested.apply(x, lambda t: _Side.apply(t, j, "bn", 2.0, None))
    (inner.sum() + outer.sum()).backward()
    assert j.pending is None
    # d/dx = 2 (nested side; returned as the nested pass' x.grad or, if it ran first, left in the join and added by the outer
    # side) + 5: in every order the sum is 7
    assert torch.equal(x.grad, torch.full((3,), 7.0)), x.grad


def test_outer_side_first_then_nested_side_raises_instead_of_guessing():
    x = torch.ones(3, requires_grad=True)
    j = GradJoin()
    inner = _Nested.apply(x, lambda t: _Side.apply(t, j, "bn", 2.0, None))
    outer = _Side.apply(x, j, "conv", 5.0, None)  # (created last: runs first, in the enclosing pass, which is still running
    with pytest.raises(RuntimeError, match="cannot be told apart"):  #  when the nested pass reaches the other side)
        (inner.sum() + outer.sum()).backward()
    assert j.pending is None
    x, j = torch.ones(3, requires_grad=True), GradJoin()
    j.put(torch.ones(3), "conv")   # left by a pass that has not completed (no final callback ran) ...
    j.task, j.done = -12345, False
    with pytest.raises(RuntimeError, match="cannot be told apart"):
        _Side.apply(x, j, "bn", 2.0, None).sum().backward()
    assert j.pending is None       # ... and nothing stale survives the error
    _Side.apply(x, j, "bn", 2.0, None).sum().backward()
    assert j.pending is not None and j.side == "bn"


def test_failed_pass_repeated_over_the_retained_graph():
    x = torch.ones(3, requires_grad=True)
    j = GradJoin()
    boom = [True]
    a = _Side.apply(x, j, "bn", 2.0, None)
    b = _Side.apply(x * 1.0, j, "conv", 5.0, boom)
    loss = a.sum() + b.sum()
    with pytest.raises(RuntimeError, match="boom"):
        loss.backward(retain_graph=True)
    stale = j.pending
    boom[0] = False
    x.grad = None
    if stale is None:     # (the failing side ran first: nothing was left)
        loss.backward()
        assert torch.equal(x.grad, torch.full((3,), 7.0))
        return
    # "bn" left its gradient in the pass that failed.  Repeated: if "bn" runs first again its stale gradient is REPLACED
    # and the sum is right; if "conv" runs first the stale one cannot be told from an enclosing pass' -> error, not a
    # silently doubled gradient
    try:
        loss.backward()
    except RuntimeError as e:
        assert "cannot be told apart" in str(e)
    else:
        assert torch.equal(x.grad, torch.full((3,), 7.0)), x.grad


def test_same_side_twice_in_one_pass_is_an_error():
    x = torch.ones(3, requires_grad=True)
    j = GradJoin()
    a = _Side.apply(x, j, "bn", 2.0, None)
    b = _Side.apply(x, j, "bn", 5.0, None)
    with pytest.raises(RuntimeError, match="twice"):
        (a.sum() + b.sum()).backward()


def test_put_outside_a_backward_pass_counts_as_completed():
    j = GradJoin()
    g = torch.ones(2)
    j.put(g, "bn")
    assert j.done and j.take("conv") is g and j.pending is None
