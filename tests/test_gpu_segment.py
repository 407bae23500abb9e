"""-m gpu: loss-side segment reductions against the numpy restatement of torch_scatter 1.4.0
(oracle/scatter.py) and, for the gradients, against torch autograd over index_add on the CPU in float64."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _labels(rng, P, S, empty=()):
    lab = rng.integers(0, S, P)
    for e in empty:
        lab[lab == e] = (e + 1) % S
    return lab.astype(np.int64)


@pytest.mark.parametrize("P,C,S", [(20000, 1, 37), (150000, 3, 120), (5000, 64, 9), (1000, 33, 700), (7, 2, 3)])
def test_scatter_add_mean_std_max(gpu, P, C, S):
    from occuseg_amd import segment
    from oracle import scatter as O
    rng = np.random.default_rng(P + C)
    src = (rng.normal(size=(P, C)) * 3 + rng.normal(size=C)).astype(np.float32)
    lab = _labels(rng, P, S, empty=(1,) if S > 3 else ())
    s, i = torch.from_numpy(src).to(gpu), torch.from_numpy(lab).to(gpu)
    assert_close(segment.scatter_add(s, i, dim=0, dim_size=S).cpu().numpy(), O.scatter_add(src, lab, S), "add")
    assert_close(segment.scatter_mean(s, i, dim=0, dim_size=S).cpu().numpy(), O.scatter_mean(src, lab, S), "mean")
    assert_close(segment.scatter_std(s, i, dim=0, dim_size=S).cpu().numpy(), O.scatter_std(src, lab, S), "std")
    assert_close(segment.scatter_std(s, i, dim=0, dim_size=S, unbiased=False).cpu().numpy(),
                 O.scatter_std(src, lab, S, unbiased=False), "std biased")
    mx, arg = segment.scatter_max(s, i, dim=0, dim_size=S)
    wmx, warg = O.scatter_max(src, lab, S)
    assert np.array_equal(mx.cpu().numpy(), wmx.astype(np.float32))  # a max is exact
    assert np.array_equal(arg.cpu().numpy(), warg)
    # dim_size inferred, 1-D source, broadcast-shaped index (as torch_scatter accepts)
    got = segment.scatter_mean(s[:, 0], i, dim=0)
    assert got.shape == (int(lab.max()) + 1,)
    assert_close(got.cpu().numpy(), O.scatter_mean(src[:, 0], lab), "mean 1-D")
    got = segment.scatter_add(s, i.view(-1, 1).expand(P, C), dim=0, dim_size=S)
    assert_close(got.cpu().numpy(), O.scatter_add(src, lab, S), "add broadcast index")


def test_scatter_gradients(gpu):
    from occuseg_amd import segment
    rng = np.random.default_rng(0)
    P, C, S = 3000, 5, 40
    src = rng.normal(size=(P, C)).astype(np.float32)
    lab = _labels(rng, P, S, empty=(3,))
    gsel = rng.normal(size=(S, C)).astype(np.float32)
    i64 = torch.from_numpy(lab)

    def ref(fn):
        x = torch.from_numpy(src).double().requires_grad_(True)
        cnt = torch.zeros(S, dtype=torch.float64).index_add_(0, i64, torch.ones(P, dtype=torch.float64))
        tot = torch.zeros(S, C, dtype=torch.float64).index_add(0, i64, x)
        if fn == "add":
            y = tot
        elif fn == "mean":
            y = tot / cnt.clamp(min=1).view(-1, 1)
        else:
            mean = tot / cnt.clamp(min=1).view(-1, 1)
            sq = torch.zeros(S, C, dtype=torch.float64).index_add(0, i64, (x - mean[i64]) ** 2)
            y = torch.sqrt(sq / (cnt - 1).clamp(min=1).view(-1, 1))
        (y * torch.from_numpy(gsel).double()).sum().backward()
        return x.grad.numpy()

    for fn in ("add", "mean", "std"):
        x = torch.from_numpy(src).to(gpu).requires_grad_(True)
        y = getattr(segment, "scatter_" + fn)(x, torch.from_numpy(lab).to(gpu), dim=0, dim_size=S)
        (y * torch.from_numpy(gsel).to(gpu)).sum().backward()
        assert_close(x.grad.cpu().numpy(), ref(fn), "grad scatter_" + fn)


def test_loss_side_usage_pattern(gpu):
    """the call pattern of examples/ScanNet/train_instance.py:224-229: per-instance mean of a norm, per-instance std"""
    from occuseg_amd import segment
    from oracle import scatter as O
    rng = np.random.default_rng(1)
    P = 40000
    lab = _labels(rng, P, 25)
    disp, disp_gt = rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, 3)).astype(np.float32)
    occ = rng.normal(size=(P, 1)).astype(np.float32)
    i = torch.from_numpy(lab).to(gpu)
    d = torch.norm(torch.from_numpy(disp).to(gpu) - torch.from_numpy(disp_gt).to(gpu), dim=1)
    got = segment.scatter_mean(d, i.view(-1), dim=0)
    want = O.scatter_mean(np.linalg.norm(disp.astype(np.float64) - disp_gt, axis=1), lab)
    assert_close(got.cpu().numpy(), want, "displacement cluster error")
    got = segment.scatter_std(torch.from_numpy(occ).to(gpu), i.view(-1), dim=0)
    assert got.shape == (25, 1)
    assert_close(got.cpu().numpy(), O.scatter_std(occ, lab), "occupancy cluster std")
    with pytest.raises(RuntimeError, match="no CPU path"):
        segment.scatter_mean(torch.zeros(4, 2), torch.zeros(4, dtype=torch.int64), dim=0)
