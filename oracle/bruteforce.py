"""Oracle cross-check: dict-based, pure-Python rule generator.  TEST INFRASTRUCTURE.

A second restatement of SURVEY.md Appendix A, written without numpy sorting or
searching so that it shares no code path with ``oracle.rules``.  Pure Python
loops: small cases only.  parity unpinned (see oracle/__init__.py).
"""


def _key(x, y, z):
    # sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:80-81
    return (((z << 21) | (y << 10) | x) & 0xFFFFFFFF) & 0x7FFFFFFF


def sites_and_input_rules(coords):
    """coords: list of (x,y,z,b), b non-decreasing.  Returns
    (site_coords [(x,y,z,b)], ctr, rule_rows [[cnt,p0,p1..]], max_active)
    per sparseconvnet/SCN/Metadata/IOLayersRules.h:135-202."""
    samples = {}
    for p, (x, y, z, b) in enumerate(coords):
        samples.setdefault(b, {}).setdefault(_key(x, y, z), []).append(p)
    site_coords, rows, ctr = [], [], [0]
    for b in range(max(samples) + 1):
        voxels = samples.get(b, {})
        for k in sorted(voxels):  # site rank = rank of key (hash_multivalue.cpp:45-93)
            pts = voxels[k]  # ascending original index (stable sort)
            x, y, z, _ = coords[pts[0]]
            site_coords.append((x, y, z, b))
            rows.append([len(pts)] + pts)
        ctr.append(len(site_coords))
    max_active = max(r[0] for r in rows)
    return site_coords, ctr, rows, max_active


def submanifold_rules(site_coords):
    """27 lists of (in, out), ascending out.  SubmanifoldRules_cuda.cu:45-114."""
    index = {c: s for s, c in enumerate(site_coords)}
    rules = [[] for _ in range(27)]
    for o, (x, y, z, b) in enumerate(site_coords):
        k = 0
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    i = index.get((x + dx, y + dy, z + dz, b))
                    if i is not None:
                        rules[k].append((i, o))
                    k += 1
    return rules


def strided_rules(site_coords):
    """(coarse_site_coords, ctr, 8 lists of (fine, coarse)).
    ConvolutionRules.h:344-378, SubmanifoldRules_cuda.cu:534-568."""
    samples = {}
    for (x, y, z, b) in site_coords:
        samples.setdefault(b, set()).add((x >> 1, y >> 1, z >> 1))
    coarse_coords, ctr, index = [], [0], {}
    nb = (max(samples) + 1) if samples else 0
    for b in range(nb):
        for (x, y, z) in sorted(samples.get(b, ()), key=lambda c: _key(*c)):
            index[(x, y, z, b)] = len(coarse_coords)
            coarse_coords.append((x, y, z, b))
        ctr.append(len(coarse_coords))
    rules = [[] for _ in range(8)]
    for t, (x, y, z, b) in enumerate(site_coords):
        k = ((x & 1) * 2 + (y & 1)) * 2 + (z & 1)
        rules[k].append((t, index[(x >> 1, y >> 1, z >> 1, b)]))
    return coarse_coords, ctr, rules
