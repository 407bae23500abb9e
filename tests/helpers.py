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
