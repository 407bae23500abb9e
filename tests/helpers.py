"""shared test helpers: seeded inputs and the comparison rule of the parity bar."""
import numpy as np
import torch

REL_TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 relative on fp32 features"


def assert_close(got, want, what="", rel=REL_TOL, scale=None):
    """|got - want| <= rel * max|want| elementwise (tolerance scaled by the tensor's
    magnitude: elements that cancel to ~0 cannot be held to a per-element ratio,
    SURVEY.md section 7 'Hard parts')."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return
    scale = max(float(np.abs(want).max()), 1e-30) if scale is None else float(scale)
    err = float(np.abs(got - want).max())
    assert np.isfinite(got).all(), what + ": non-finite values"
    assert err <= rel * scale, "%s: max abs err %.3e > %.1e * max|ref| %.3e" % (what, err, rel, scale)


def random_points(seed, n_per_sample, batch, extent, origin=0):
    """duplicated random voxels, batch column non-decreasing; int64 [P,4]"""
    rng = np.random.default_rng(seed)
    out = []
    for b in range(batch):
        c = rng.integers(0, extent, (n_per_sample, 3)) + origin
        out.append(np.concatenate([c, np.full((n_per_sample, 1), b)], 1))
    return np.concatenate(out).astype(np.int64)


def build_metadata(coords, device, mode=4, spatial=4096):
    """run the product InputLayer path on coords (numpy int64 [P,3|4]); returns (metadata, spatial_size)"""
    from occuseg_amd import SCN
    m = SCN.Metadata_3()
    size = torch.tensor([spatial] * 3, dtype=torch.int64)
    m.inputLayer(size, torch.from_numpy(np.ascontiguousarray(coords)), 0, mode, device)
    return m, size


def device_masks(net, scn, run):
    """run() one forward pass of `net` on the device with a hook on every batch norm; returns run()'s result and
    {id(module): sign of the module's output as the DEVICE computed it}.  An element with |y| ~ 1e-7 has a different sign in
    fp32 and fp64; handed to oracle.torch_ref.ref_forward(masks=...), the oracle's backward pass then differentiates the same
    piecewise-linear function as the device and every gradient can be held to 1e-4 (without it one flipped mask moves a summed
    gradient by a whole term: 1.2e-3 of a tensor's magnitude observed for one batch-norm gamma)."""
    masks, hooks = {}, []
    for mod in net.modules():
        if isinstance(mod, scn.BatchNormalization):
            hooks.append(mod.register_forward_hook(
                lambda md, inp, out: masks.__setitem__(id(md), (out.features.detach() > 0).cpu())))
    try:
        res = run()
    finally:
        for h in hooks:
            h.remove()
    return res, masks


def grads_vs_oracle(net, stats, rel=REL_TOL):
    """every parameter gradient of `net` (and the batch norms' running statistics) against oracle.torch_ref's `stats`"""
    for mod in net.modules():
        ref = stats["params"].get(id(mod))
        if not ref:
            continue
        for name, leaf in ref.items():
            got = getattr(mod, name).grad
            assert got is not None, (mod, name)
            assert_close(got.cpu().numpy(), leaf.grad.numpy(), "grad %s.%s" % (mod, name), rel=rel)
        if id(mod) in stats["bn"]:
            mean, var = stats["bn"][id(mod)]
            assert_close(mod.running_mean.cpu().numpy(), (0.1 * mean).numpy(), "running_mean")
            if var is not None:
                assert_close(mod.running_var.cpu().numpy(), (0.9 + 0.1 * var).numpy(), "running_var")
