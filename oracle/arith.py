"""Oracle: arithmetic of the reference's CPU path (numpy restatement).  TEST INFRASTRUCTURE.

Every function takes ``dtype``: np.float32 reproduces the reference's precision,
np.float64 gives the tight comparison target used for the 1e-4 parity bar
("Hard parts": compare against a float64 oracle with
tolerance scaled by max|x|).  File references are relative to /root/reference/.

Pinned against the reference's own loops compiled into oracle/_ref (see
oracle/Makefile, tests/test_oracle_ref.py and tests/golden/).
"""
import numpy as np


def _f(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def conv_forward(inp, weight, rules, n_out, bias=None, in_col=0, out_col=1, dtype=np.float32):
    """out[o,:] = bias + sum_k sum_{(i,o) in rules[k]} in[i,:] @ W[k]

    sparseconvnet/SCN/CPU/Convolution.cpp:35-75 (strided), :114-149 (submanifold):
    rule_index_select -> at::mm -> rule_index_add_.  Deconvolution is the same
    with in_col=1, out_col=0 (sparseconvnet/SCN/CPU/Deconvolution.cpp:37-40).
    Returns (out, multiply_adds) -- the double returned across the boundary.
    """
    inp, weight = _f(inp, dtype), _f(weight, dtype)
    out = np.zeros((n_out, weight.shape[2]), dtype)
    if bias is not None and np.size(bias) and n_out:
        out[:] = _f(bias, dtype)
    flops = 0.0
    for k, r in enumerate(rules):
        if len(r) == 0:
            continue
        flops += float(len(r)) * weight.shape[1] * weight.shape[2]
        rows = inp[r[:, in_col]] @ weight[k]
        np.add.at(out, r[:, out_col], rows)
    return out, flops


def conv_backward(inp, d_out, weight, rules, in_col=0, out_col=1, want_bias=False, dtype=np.float32):
    """d_in[i,:] += d_out[o,:] @ W[k]^T ; dW[k] = in[rules.in]^T @ d_out[rules.out] ;
    d_bias = column sums of d_out.
    sparseconvnet/SCN/CPU/Convolution.cpp:77-112, :151-189; Deconvolution.cpp:47-84."""
    inp, d_out, weight = _f(inp, dtype), _f(d_out, dtype), _f(weight, dtype)
    d_in = np.zeros_like(inp)
    d_w = np.zeros_like(weight)
    d_b = d_out.sum(0) if want_bias else None
    for k, r in enumerate(rules):
        if len(r) == 0:
            continue
        rows_in = inp[r[:, in_col]]
        rows_do = d_out[r[:, out_col]]
        d_w[k] = rows_in.T @ rows_do
        np.add.at(d_in, r[:, in_col], rows_do @ weight[k].T)
    return d_in, d_w, d_b


def batchnorm_forward(x, weight, bias, running_mean, running_var, eps=1e-4, momentum=0.9,
                      train=True, leakiness=1.0, dtype=np.float32):
    """sparseconvnet/SCN/CPU/BatchNormalization.cpp:12-62.

    train: mean = sum(x)/N ; s = sum(x^2) - mean^2 N ; running_mean = mom*rm + (1-mom)*mean ;
    running_var = mom*rv + (1-mom)*s/(N-1) ; invstd = (s/N + eps)^-1/2.
    y = leaky(x*w' + b'), w' = invstd*gamma, b' = beta - mean*w'.
    Returns (y, save_mean, save_invstd, new_running_mean, new_running_var).
    """
    x = _f(x, dtype)
    n, c = x.shape
    rm, rv = _f(running_mean, dtype).copy(), _f(running_var, dtype).copy()
    if train:
        mean = x.sum(0, dtype=dtype) / dtype(n)
        s = (x * x).sum(0, dtype=dtype) - mean * mean * dtype(n)
        rm = dtype(momentum) * rm + dtype(1 - momentum) * mean
        with np.errstate(divide="ignore", invalid="ignore"):
            rv = dtype(momentum) * rv + dtype(1 - momentum) * s / dtype(n - 1)
        invstd = (s / dtype(n) + dtype(eps)) ** dtype(-0.5)
    else:
        mean = rm.copy()
        invstd = (rv + dtype(eps)) ** dtype(-0.5)
    g = _f(weight, dtype) if weight is not None and np.size(weight) else np.ones(c, dtype)
    b = _f(bias, dtype) if bias is not None and np.size(bias) else np.zeros(c, dtype)
    w = invstd * g
    bb = -mean * w + b
    y = x * w + bb
    y = np.where(y > 0, y, y * dtype(leakiness)).astype(dtype)
    return y, mean.astype(dtype), invstd.astype(dtype), rm.astype(dtype), rv.astype(dtype)


def batchnorm_backward(x, y, d_out, save_mean, save_invstd, weight, leakiness=1.0, dtype=np.float32):
    """sparseconvnet/SCN/CPU/BatchNormalization.cpp:64-107.
    Returns (d_in, d_out_masked, d_weight, d_bias); d_out_masked is what the
    reference writes back INTO d_output_features (:79-82)."""
    x, y, d_out = _f(x, dtype), _f(y, dtype), _f(d_out, dtype)
    mean, invstd = _f(save_mean, dtype), _f(save_invstd, dtype)
    n, c = x.shape
    d = d_out * np.where(y > 0, dtype(1), dtype(leakiness))
    grad_sum = d.sum(0, dtype=dtype)
    dotp = ((x - mean) * d).sum(0, dtype=dtype)
    grad_mean = grad_sum / dtype(n)
    k = dotp * invstd * invstd / dtype(n)
    g = _f(weight, dtype) if weight is not None and np.size(weight) else np.ones(c, dtype)
    d_in = (d - grad_mean - (x - mean) * k) * invstd * g
    return d_in.astype(dtype), d.astype(dtype), (dotp * invstd).astype(dtype), grad_sum.astype(dtype)


def input_layer_forward(feats, rules, mode=4, dtype=np.float32):
    """out[s] = sum_j mult * in[p_j], mult = 1/cnt for mode 4.
    sparseconvnet/SCN/CPU/IOLayers.cpp:12-30 (== CUDA/IOLayers.cu:17-31)."""
    feats = _f(feats, dtype)
    n = rules.shape[0]
    out = np.zeros((n, feats.shape[1]), dtype)
    cnt = rules[:, 0]
    mult = np.where((mode == 4) & (cnt > 0), dtype(1) / np.maximum(cnt, 1).astype(dtype), dtype(1)).astype(dtype)
    for j in range(rules.shape[1] - 1):
        m = cnt > j
        out[m] += mult[m, None] * feats[rules[m, 1 + j]]
    return out


def input_layer_backward(d_out, rules, n_points, mode=4, dtype=np.float32):
    """d_in[p_j] += mult * d_out[s].  sparseconvnet/SCN/CPU/IOLayers.cpp:31-47."""
    d_out = _f(d_out, dtype)
    d_in = np.zeros((n_points, d_out.shape[1]), dtype)
    cnt = rules[:, 0]
    mult = np.where((mode == 4) & (cnt > 0), dtype(1) / np.maximum(cnt, 1).astype(dtype), dtype(1)).astype(dtype)
    for j in range(rules.shape[1] - 1):
        m = cnt > j
        d_in[rules[m, 1 + j]] += mult[m, None] * d_out[m]
    return d_in


def output_layer_forward(feats, rules, n_points, dtype=np.float32):
    """out[p_j] = in[s] (no averaging).  sparseconvnet/SCN/CPU/IOLayers.cpp:94-112."""
    return input_layer_backward(feats, rules, n_points, mode=3, dtype=dtype)


def output_layer_backward(d_out, rules, dtype=np.float32):
    """d_in[s] = sum_j d_out[p_j].  sparseconvnet/SCN/CPU/IOLayers.cpp:113-132."""
    return input_layer_forward(d_out, rules, mode=3, dtype=dtype)


def nin_forward(x, weight, bias=None, dtype=np.float32):
    """sparseconvnet/SCN/CPU/NetworkInNetwork.cpp:8-24 (addmm)."""
    out = _f(x, dtype) @ _f(weight, dtype)
    if bias is not None and np.size(bias):
        out = out + _f(bias, dtype)
    return out


def nin_backward(x, d_out, weight, dtype=np.float32):
    """sparseconvnet/SCN/CPU/NetworkInNetwork.cpp:25-46."""
    x, d_out, weight = _f(x, dtype), _f(d_out, dtype), _f(weight, dtype)
    return d_out @ weight.T, x.T @ d_out, d_out.sum(0)
