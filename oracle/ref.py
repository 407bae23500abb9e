"""ctypes view of oracle/_ref/libscn_ref.so -- the reference's own CPU loops
(sparseconvnet/SCN/CPU/{BatchNormalization,IOLayers,Convolution}.cpp) compiled by oracle/Makefile.
TEST INFRASTRUCTURE: used to pin oracle.arith and to generate tests/golden/*.npz."""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (libscn_ref.so links libtorch_cpu)

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libscn_ref.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
    return _lib


# precision of the reference's templates this module calls: float32 (what the reference runs; entry points without suffix) or
# float64 (the same templates instantiated at double, suffix _f64: the reference's own loops without their float32 rounding)
_DT = [np.float32]


class precision(object):
    """with ref.precision(np.float64): ... -- every call inside goes to the float64 instantiation"""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype).type

    def __enter__(self):
        self.before = _DT[0]
        _DT[0] = self.dtype

    def __exit__(self, *exc):
        _DT[0] = self.before


def _f(a):
    return np.ascontiguousarray(a, dtype=_DT[0])


def _fn(name):
    return getattr(_load(), name + ("_f64" if _DT[0] is np.float64 else ""))


def _c(v):
    return ctypes.c_double(v) if _DT[0] is np.float64 else ctypes.c_float(v)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def bn_forward(x, weight, bias, running_mean, running_var, eps=1e-4, momentum=0.9, train=True, leakiness=1.0):
    x = _f(x)
    n, c = x.shape
    y = np.empty_like(x)
    sm, si = np.zeros(c, _DT[0]), np.zeros(c, _DT[0])
    rm, rv = _f(running_mean).copy(), _f(running_var).copy()
    w = None if weight is None else _f(weight)
    b = None if bias is None else _f(bias)
    _fn("ref_bn_forward")(_p(x), _p(y), ctypes.c_int32(c), ctypes.c_int32(n), _p(sm), _p(si), _p(rm), _p(rv), _p(w),
                           _p(b), _c(eps), _c(momentum), ctypes.c_int(int(train)),
                           _c(leakiness))
    return y, sm, si, rm, rv


def bn_backward(x, y, d_out, save_mean, save_invstd, weight, leakiness=1.0):
    x, y, d = _f(x), _f(y), _f(d_out).copy()
    n, c = x.shape
    dx = np.empty_like(x)
    w = None if weight is None else _f(weight)
    dw, db = np.zeros(c, _DT[0]), np.zeros(c, _DT[0])
    sm, si = _f(save_mean), _f(save_invstd)
    dummy = np.zeros(c, _DT[0])
    _fn("ref_bn_backward")(_p(x), _p(dx), _p(y), _p(d), ctypes.c_int32(c), ctypes.c_int32(n), _p(sm), _p(si), _p(dummy),
                            _p(dummy), _p(w), _p(dummy), _p(dw), _p(db), _c(leakiness))
    return dx, d, dw, db


def input_forward(feats, table, mode=4):
    feats = _f(feats)
    table = np.ascontiguousarray(table, np.int32)
    n, wdt = table.shape
    out = np.zeros((n, feats.shape[1]), _DT[0])
    _fn("ref_input_forward")(_p(feats), _p(out), ctypes.c_int32(n), ctypes.c_int32(wdt - 1),
                              ctypes.c_int32(feats.shape[1]), _p(table), ctypes.c_int(int(mode == 4)))
    return out


def input_backward(d_out, table, n_points, mode=4):
    d_out = _f(d_out)
    table = np.ascontiguousarray(table, np.int32)
    n, wdt = table.shape
    d_in = np.zeros((n_points, d_out.shape[1]), _DT[0])
    _fn("ref_input_backward")(_p(d_in), _p(d_out), ctypes.c_int32(n), ctypes.c_int32(wdt - 1),
                               ctypes.c_int32(d_out.shape[1]), _p(table), ctypes.c_int(int(mode == 4)))
    return d_in


def conv_forward(inp, weight, rules, n_out, in_col=0):
    """per offset: rule_index_select -> at::mm -> rule_index_add_ (reference helpers)"""
    inp, weight = _f(inp), _f(weight)
    out = np.zeros((n_out, weight.shape[2]), _DT[0])
    for k, r in enumerate(rules):
        r = np.ascontiguousarray(r, np.int32)
        wk = np.ascontiguousarray(weight[k])
        _fn("ref_rule_conv_offset")(_p(inp), ctypes.c_int64(inp.shape[0]), _p(out), ctypes.c_int64(n_out), _p(wk),
                                     ctypes.c_int32(weight.shape[1]), ctypes.c_int32(weight.shape[2]), _p(r),
                                     ctypes.c_int32(len(r)), ctypes.c_int(in_col))
    return out


def conv_backward(inp, d_out, weight, rules, in_col=0):
    inp, d_out, weight = _f(inp), _f(d_out), _f(weight)
    d_in = np.zeros_like(inp)
    d_w = np.zeros_like(weight)
    for k, r in enumerate(rules):
        r = np.ascontiguousarray(r, np.int32)
        wk = np.ascontiguousarray(weight[k])
        dwk = np.zeros_like(wk)
        _fn("ref_rule_conv_offset_backward")(_p(inp), _p(d_in), ctypes.c_int64(inp.shape[0]), _p(d_out),
                                              ctypes.c_int64(d_out.shape[0]), _p(wk), _p(dwk),
                                              ctypes.c_int32(weight.shape[1]), ctypes.c_int32(weight.shape[2]), _p(r),
                                              ctypes.c_int32(len(r)), ctypes.c_int(in_col))
        d_w[k] = dwk
    return d_in, d_w
