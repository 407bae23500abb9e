"""Synthetic ScanNet-like scenes (ScanNet itself is not available offline).

Follows the recipe of SURVEY.md section 8(d): a room shell (floor + 4 walls, no
ceiling) with random axis-aligned boxes standing on the floor and a few tilted
planes, surface points sampled uniformly, voxelised at ``scale`` voxels per
metre (reference default 50 = 2 cm, examples/ScanNet/config.py), shifted so the
minimum coordinate is 10 (examples/ScanNet/datasets/scannet.py:133), duplicates
kept so that InputLayer mode 4 averages them, colours U[-1,1]^3.

Also the two single-layer stress site sets of BASELINE.json configs[4].
"""
import numpy as np


def _rect(rng, n, origin, u, v):
    a = rng.random((n, 1))
    b = rng.random((n, 1))
    return origin + a * u + b * v


def _surfaces(rng, s):
    """list of (origin, u, v) rectangles in metres; s scales the whole scene."""
    L, W, H = 8.0 * s, 6.0 * s, 3.0 * s
    rects = [
        (np.array([0, 0, 0.0]), np.array([L, 0, 0]), np.array([0, W, 0])),  # floor
        (np.array([0, 0, 0.0]), np.array([L, 0, 0]), np.array([0, 0, H])),  # wall y=0
        (np.array([0, W, 0.0]), np.array([L, 0, 0]), np.array([0, 0, H])),  # wall y=W
        (np.array([0, 0, 0.0]), np.array([0, W, 0]), np.array([0, 0, H])),  # wall x=0
        (np.array([L, 0, 0.0]), np.array([0, W, 0]), np.array([0, 0, H])),  # wall x=L
    ]
    for _ in range(12):  # boxes on the floor: 4 sides + top
        e = rng.uniform(0.3, 2.0, 3) * s
        e[2] = min(e[2], H * 0.9)
        o = np.array([rng.uniform(0, L - e[0]), rng.uniform(0, W - e[1]), 0.0])
        ex, ey, ez = np.array([e[0], 0, 0]), np.array([0, e[1], 0]), np.array([0, 0, e[2]])
        rects += [(o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez), (o + ez, ex, ey)]
    for _ in range(4):  # tilted 1x1 m planes
        o = np.array([rng.uniform(0.5 * s, L - 1.5 * s), rng.uniform(0.5 * s, W - 1.5 * s), rng.uniform(0.3 * s, H - 1.2 * s)])
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        b = np.cross(a, rng.normal(size=3))
        b /= np.linalg.norm(b)
        rects.append((o, a * s, b * s))
    return rects


def scene_points(seed, target_sites=150_000, scale=50, pts_per_voxel=2.5, noise_vox=0.35):
    """One scene.  Returns (coords int64 [P,3], colors float32 [P,3]).

    The scene scale factor is chosen so that (surface area / voxel area) is about
    ``target_sites``; ``pts_per_voxel`` points are sampled per surface voxel so
    that most surface voxels are hit and many hold duplicates.
    """
    rng = np.random.default_rng(seed)
    rects0 = _surfaces(np.random.default_rng(seed), 1.0)
    area0 = sum(np.linalg.norm(np.cross(u, v)) for _, u, v in rects0)
    vox_area = (1.0 / scale) ** 2
    # effective thickness/occupancy factor ~1.34 found empirically for noise_vox=0.35, 2.5 pts/voxel
    s = np.sqrt(target_sites * vox_area / (1.34 * area0))
    rects = _surfaces(np.random.default_rng(seed), s)
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in rects])
    n_total = int(pts_per_voxel * areas.sum() / vox_area)
    counts = rng.multinomial(n_total, areas / areas.sum())
    pts = []
    for (o, u, v), n in zip(rects, counts):
        p = _rect(rng, n, o, u, v)
        nrm = np.cross(u, v)
        nrm /= np.linalg.norm(nrm)
        p = p + nrm * rng.normal(0, noise_vox / scale, (n, 1))
        pts.append(p)
    pts = np.concatenate(pts)
    vox = np.floor(pts * scale).astype(np.int64)
    vox -= vox.min(0)
    vox += 10
    assert vox[:, 0].max() < 1024 and vox[:, 1].max() < 2048 and vox[:, 2].max() < 1024
    perm = rng.permutation(len(vox))  # points arrive in scan order, not sorted
    colors = rng.uniform(-1, 1, (len(vox), 3)).astype(np.float32)
    return vox[perm], colors


def batch(seeds, target_sites=150_000, scale=50):
    """Merge scenes the way ScanNet.trainMerge does
    (examples/ScanNet/datasets/scannet.py:254-265): coords [P,4] with the batch
    index in the last column, samples concatenated in order.
    Returns (coords int64 [P,4], feats float32 [P,3])."""
    cs, fs = [], []
    for b, seed in enumerate(seeds):
        c, f = scene_points(seed, target_sites, scale)
        cs.append(np.concatenate([c, np.full((len(c), 1), b, np.int64)], 1))
        fs.append(f)
    return np.concatenate(cs), np.concatenate(fs)


def bernoulli_grid(side, p, seed, offset=0):
    """side^3 box with Bernoulli(p) occupancy (configs[0]: side 32, p .25, seed 0;
    configs[4] 5a: side 160, p .488, seed 7).  Returns coords int64 [N,3]."""
    rng = np.random.default_rng(seed)
    occ = rng.random((side, side, side)) < p
    x, y, z = np.nonzero(occ)
    return np.stack([x, y, z], 1).astype(np.int64) + offset


def solid_block(side, offset=0):
    """Solid side^3 block (configs[4] 5b: side 126 -> 2,000,376 sites)."""
    r = np.arange(side, dtype=np.int64)
    x, y, z = np.meshgrid(r, r, r, indexing="ij")
    return np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + offset
