"""Oracle: segment reductions over labels (numpy, float64).  TEST INFRASTRUCTURE.

torch_scatter is a third-party dependency of the reference (p1.yml:167 pins torch-scatter==1.4.0) and is not part of
/root/reference nor installed here; its published algorithm for the four functions the reference's losses call
(examples/ScanNet/train_instance.py:224-229, examples/ScanNet/discriminative.py:52-53,146) is restated:

    scatter_add(src, index, dim=0)[s]  = sum_{p: index[p]=s} src[p]
    scatter_mean                        = scatter_add / clamp(count, min=1)
    scatter_std(unbiased=True)          = sqrt( scatter_add((src - mean[index])^2) / clamp(count - 1, min=1) )
    scatter_max                         = (max over the segment, argmax row); empty segments: (0, P)

(torch_scatter/{add,mean,std,max}.py of release 1.4.0.)  parity unpinned by reference outputs: the reference holds no
tests or fixtures for its losses; the call sites fix dim = 0 and the [points, planes] layout.
"""
import numpy as np


def scatter_add(src, index, dim_size=None):
    src = np.asarray(src, np.float64)
    S = int(dim_size) if dim_size is not None else int(index.max()) + 1
    out = np.zeros((S,) + src.shape[1:], np.float64)
    np.add.at(out, index, src)
    return out


def counts(index, S):
    return np.bincount(index, minlength=S).astype(np.float64)


def scatter_mean(src, index, dim_size=None):
    tot = scatter_add(src, index, dim_size)
    c = np.maximum(counts(index, tot.shape[0]), 1)
    return tot / c.reshape((-1,) + (1,) * (tot.ndim - 1))


def scatter_std(src, index, dim_size=None, unbiased=True):
    src = np.asarray(src, np.float64)
    mean = scatter_mean(src, index, dim_size)
    sq = scatter_add((src - mean[index]) ** 2, index, mean.shape[0])
    c = counts(index, mean.shape[0])
    denom = np.maximum(c - 1 if unbiased else c, 1)
    return np.sqrt(sq / denom.reshape((-1,) + (1,) * (sq.ndim - 1)))


def scatter_max(src, index, dim_size=None):
    src = np.asarray(src, np.float64)
    S = int(dim_size) if dim_size is not None else int(index.max()) + 1
    P = src.shape[0]
    mx = np.zeros((S,) + src.shape[1:], np.float64)
    arg = np.full((S,) + src.shape[1:], P, np.int64)
    for s in range(S):
        rows = np.nonzero(index == s)[0]
        if len(rows):
            sub = src[rows]
            mx[s] = sub.max(0)
            arg[s] = rows[sub.argmax(0)]
    return mx, arg
