"""Oracle: site ordering and rulebooks (numpy restatement).  TEST INFRASTRUCTURE.

Restates the *GPU path* of the reference, the only one OccuSeg runs
(``#define GPU_GRID`` in sparseconvnet/SCN/Metadata/Metadata.h:42).  All file
references are relative to /root/reference/.

parity unpinned: the reference holds no golden vectors for these functions and
its CUDA/CUDPP code cannot run here; see oracle/__init__.py.
"""
import numpy as np

KEY_MASK = 0x7FFFFFFF
# valid coordinate box of the 31-bit key (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:80-83)
BOX = (1024, 2048, 1024)  # x < 2^10, y < 2^11, z < 2^10


def key31(x, y, z):
    """key = ((z<<21) | (y<<10) | x) & 0x7FFFFFFF
    (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:80-81 and :113-114)."""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    z = np.asarray(z, dtype=np.int64)
    k = ((z << 21) | (y << 10) | x) & 0xFFFFFFFF  # assignment to uint32_t
    return (k & KEY_MASK).astype(np.int64)


def in_box(x, y, z):
    return ((x >= 0) & (x < BOX[0]) & (y >= 0) & (y < BOX[1]) &
            (z >= 0) & (z < BOX[2]))


class Grid:
    """Active sites of one scale: the oracle's view of GPU_SparseGrids
    (sparseconvnet/SCN/Metadata/Metadata.h:114-120).

    coords : int64 [N,4]  (x,y,z,b) in global site order
    ctr    : int64 [B+1]  running offsets, ctr[b] = first global id of sample b
             (sparseconvnet/SCN/Metadata/IOLayersRules.h:164-169)
    gkey   : int64 [N]    (b<<31 | key31) -- ascending by construction
    """

    def __init__(self, coords, ctr):
        self.coords = np.ascontiguousarray(coords, dtype=np.int64)
        self.ctr = np.asarray(ctr, dtype=np.int64)
        c = self.coords
        self.gkey = (c[:, 3] << 31) | key31(c[:, 0], c[:, 1], c[:, 2])
        assert np.all(np.diff(self.gkey) > 0), "sites must be strictly key-ascending per sample"

    @property
    def n(self):
        return self.coords.shape[0]

    @property
    def batch(self):
        return len(self.ctr) - 1

    def lookup(self, x, y, z, b):
        """site id of voxel (x,y,z) in sample b, or -1.  Out-of-box queries are
        inactive (explicit bounds test instead of the reference's
        aliasing all-ones key, sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:67-69)."""
        ok = in_box(x, y, z)
        q = (np.asarray(b, np.int64) << 31) | key31(np.where(ok, x, 0), np.where(ok, y, 0), np.where(ok, z, 0))
        pos = np.searchsorted(self.gkey, q)
        pos_c = np.minimum(pos, max(self.n - 1, 0))
        hit = ok & (pos < self.n)
        if self.n:
            hit &= self.gkey[pos_c] == q
        else:
            hit &= False
        return np.where(hit, pos_c, -1).astype(np.int64)


def input_layer_rules(coords, batch_size=None, mode=4):
    """Input layer: voxelise points, number the sites, emit input rules.

    Follows inputLayerRulesSimple (sparseconvnet/SCN/Metadata/IOLayersRules.h:135-202):
      * coords int64 [P,4] = (x,y,z,b), b non-decreasing
        (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:72-100 detects sample starts by b[i]<b[i+1]);
        [P,3] means a single sample.
      * per sample: stable radix sort of (key, global row index) pairs, unique
        keys, site rank = index of key among the unique keys
        (extra/cudpp/src/cudpp_hash/hash_multivalue.cpp:45-93);
      * site coords = coords of the first duplicate in sorted order
        (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:224-226);
      * rule row s = [cnt, p_0 < p_1 < ...] padded to 1+maxActive
        (sparseconvnet/SCN/CUDA/CUDPPWrapper.cu:53-64).  Padding is written as 0 here.

    Returns (grid, header, rules, site_of_point):
      header = [mode, maxActive, P, N] (IOLayersRules.h:192-199),
      rules int32 [N, 1+maxActive], site_of_point int64 [P] (``_d_hash_idx``).
    """
    coords = np.asarray(coords, dtype=np.int64)
    assert mode in (3, 4), "GPU input layer asserts mode 3/4 (IOLayersRules.h:141)"
    if coords.shape[1] == 3:
        coords = np.concatenate([coords, np.zeros((coords.shape[0], 1), np.int64)], 1)
    P = coords.shape[0]
    b = coords[:, 3]
    assert P > 0 and np.all(np.diff(b) >= 0), "batch column must be non-decreasing"
    B = int(b[-1]) + 1
    if batch_size is not None:
        B = max(B, int(batch_size))
    assert len(np.unique(b)) == int(b[-1]) + 1 and b[0] == 0, "every sample must be non-empty (SURVEY A.2)"
    k = key31(coords[:, 0], coords[:, 1], coords[:, 2])
    g = (b << 31) | k
    order = np.argsort(g, kind="stable")  # == per-sample stable sort, samples in batch order
    gs = g[order]
    first = np.ones(P, bool)
    first[1:] = gs[1:] != gs[:-1]
    site_sorted = np.cumsum(first) - 1  # global site id of each sorted point
    N = int(site_sorted[-1]) + 1
    site_of_point = np.empty(P, np.int64)
    site_of_point[order] = site_sorted
    site_coords = coords[order[first]]
    counts = np.bincount(site_sorted, minlength=N)
    max_active = int(counts.max())
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rules = np.zeros((N, 1 + max_active), np.int32)
    rules[:, 0] = counts
    j = np.arange(P) - start[site_sorted]
    rules[site_sorted, 1 + j] = order  # ascending original index within a site (stable sort)
    per_sample = np.bincount(site_coords[:, 3], minlength=B)
    ctr = np.concatenate([[0], np.cumsum(per_sample)])
    grid = Grid(site_coords, ctr)
    header = np.array([mode, max_active, P, N], np.int32)
    return grid, header, rules, site_of_point


def submanifold_rules(grid):
    """3x3x3 submanifold rulebook.

    k = (dx+1)*9 + (dy+1)*3 + (dz+1)  (x outer, z inner:
    sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:61-74; same numbering as
    RectangularRegion::offset, sparseconvnet/SCN/Metadata/RectangularRegions.h:52-60).
    For each output site o in ascending global id, pair (in, out) = (site at
    coord(o)+(dx,dy,dz), o) if that voxel is active in the same sample
    (SubmanifoldRules_cuda.cu:91-114; order-preserving compaction
    sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cpp:167-176; samples appended in
    batch order, sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:450-453).

    Returns (rules, nbr): rules = list of 27 int32 [R_k,2]; nbr = int32 [N,27]
    dense neighbour table (-1 = inactive), the ``d_global_query_result`` of the
    reference (SubmanifoldRules_cuda.cu:91-104) with ctr already added.
    """
    c = grid.coords
    N = grid.n
    nbr = np.full((N, 27), -1, np.int32)
    rules = []
    out_ids = np.arange(N, dtype=np.int64)
    k = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                i = grid.lookup(c[:, 0] + dx, c[:, 1] + dy, c[:, 2] + dz, c[:, 3])
                nbr[:, k] = i
                m = i >= 0
                rules.append(np.stack([i[m], out_ids[m]], 1).astype(np.int32))
                k += 1
    return rules, nbr


def strided_rules(grid):
    """size=2, stride=2 convolution rulebook and the coarse grid.

    Follows Convolution_InputSgToRulesAndOutputSg_FastDownSampleMode
    (sparseconvnet/SCN/Metadata/ConvolutionRules.h:344-378): coarse coord = fine/2
    (:370); coarse sites numbered per sample by key rank with their own running
    ctr (:372-378, driver :770-828); offset k = ((x&1)*2 + (y&1))*2 + (z&1)
    (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:549-554); pairs
    (fine, coarse) appended in ascending fine id (:562-564, compaction :603-613).

    Returns (coarse_grid, rules, child): rules = list of 8 int32 [R_k,2] with
    column 0 = fine site (conv input), column 1 = coarse site (conv output);
    child = int32 [N_coarse, 8] inverse table (fine site or -1).
    """
    c = grid.coords
    cc = c.copy()
    cc[:, :3] = c[:, :3] >> 1
    g = (cc[:, 3] << 31) | key31(cc[:, 0], cc[:, 1], cc[:, 2])
    ug, first_idx, inv = np.unique(g, return_index=True, return_inverse=True)
    coarse_coords = cc[first_idx]
    per_sample = np.bincount(coarse_coords[:, 3], minlength=grid.batch)
    ctr = np.concatenate([[0], np.cumsum(per_sample)])
    coarse = Grid(coarse_coords, ctr)
    off = ((c[:, 0] & 1) * 2 + (c[:, 1] & 1)) * 2 + (c[:, 2] & 1)
    fine_ids = np.arange(grid.n, dtype=np.int64)
    rules = []
    child = np.full((coarse.n, 8), -1, np.int32)
    for k in range(8):
        m = off == k
        rules.append(np.stack([fine_ids[m], inv[m]], 1).astype(np.int32))
        child[inv[m], k] = fine_ids[m]
    return coarse, rules, child


# ------------------------------------------------------- normal-guided filters ---
# rot_index (sparseconvnet/SCN/CUDA/SubmanifoldRules_cuda.cu:8-15) = the `index` table of remap_rules_with_normal
# (sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:217-223): rows = orientation classes 0..5 (only 0, 2, 4 are
# ever produced by OrientedFilter), columns = original offset, entry = list the pair moves to.
ROT_INDEX = np.array([
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
    24, 25, 26, 21, 22, 23, 18, 19, 20, 15, 16, 17, 12, 13, 14, 9, 10, 11, 6, 7, 8, 3, 4, 5, 0, 1, 2,
    6, 7, 8, 15, 16, 17, 24, 25, 26, 3, 4, 5, 12, 13, 14, 21, 22, 23, 0, 1, 2, 9, 10, 11, 18, 19, 20,
    18, 19, 20, 9, 10, 11, 0, 1, 2, 21, 22, 23, 12, 13, 14, 3, 4, 5, 24, 25, 26, 15, 16, 17, 6, 7, 8,
    2, 11, 20, 5, 14, 23, 8, 17, 26, 1, 10, 19, 4, 13, 22, 7, 16, 25, 0, 9, 18, 3, 12, 21, 6, 15, 24,
    18, 9, 0, 21, 12, 3, 24, 15, 6, 19, 10, 1, 22, 13, 4, 25, 16, 7, 20, 11, 2, 23, 14, 5, 26, 17, 8], np.int64).reshape(6, 27)


def site_normals(point_normals, table):
    """per-site normal = normalize(mean of the normals of the site's input points), float32 arithmetic in the
    reference's order (sparseconvnet/SCN/CUDA/IOLayers.cpp:43-62; Float3::normalize, sparseconvnet/SCN/Metadata/
    Metadata.h:95-101: left as it is when its length is < 1e-8).  table = the input rule table [N, 1+maxActive]."""
    pn = np.asarray(point_normals, np.float32)
    N = table.shape[0]
    out = np.zeros((N, 3), np.float32)
    for s in range(N):
        cnt = int(table[s, 0])
        acc = np.zeros(3, np.float32)
        for j in range(cnt):
            acc = (acc + pn[table[s, 1 + j]]).astype(np.float32)
        if cnt > 0:
            acc = (acc / np.float32(cnt)).astype(np.float32)
        mag = np.sqrt(np.float32(acc[0] * acc[0]) + np.float32(acc[1] * acc[1]) + np.float32(acc[2] * acc[2]), dtype=np.float32)
        if not mag < 1e-8:
            acc = (acc * (np.float32(1) / mag)).astype(np.float32)
        out[s] = acc
    return out


def oriented_filter(normals):
    """OrientedFilter (sparseconvnet/SCN/Metadata/RectangularRegions.h:12-32): 0 when |x| is the largest component
    (ties first), 2 for |y|, 4 for |z|"""
    a = np.abs(np.asarray(normals, np.float32))
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    face = np.zeros(len(a), np.int64)
    is_x = (x >= y) & (x >= z)
    is_y = ~is_x & (y >= x) & (y >= z)
    is_z = ~is_x & ~is_y & (z >= x) & (z >= y)
    face[is_y] = 2
    face[is_z] = 4
    return face


def remap_rules_with_normal(rules, ori):
    """remap_rules_with_normal (sparseconvnet/SCN/Metadata/SubmanifoldConvolutionRules.h:213-245): walk the old lists
    i = 0..26 in order, every pair (in, out) is appended to list ROT_INDEX[ori[out]][i]."""
    new = [[] for _ in range(27)]
    for i in range(27):
        r = rules[i]
        if len(r):
            tgt = ROT_INDEX[ori[r[:, 1]], i]
            for n in np.unique(tgt):
                new[n].append(r[tgt == n])
    return [np.concatenate(x).astype(np.int32) if x else np.zeros((0, 2), np.int32) for x in new]


def resolution_based_scattering(points_lr, points_hr, stride):
    """ResolutionBasedScatteringCuda (sparseconvnet/SCN/Metadata/ConvolutionRules.h:327-342) for one sample: rank of
    key(points_hr / stride) among the unique sorted keys of points_lr, or -1."""
    lr = np.asarray(points_lr, np.int64)
    hr = np.asarray(points_hr, np.int64) // int(stride)
    keys = np.unique(key31(lr[:, 0], lr[:, 1], lr[:, 2]))
    q = key31(hr[:, 0], hr[:, 1], hr[:, 2])
    pos = np.searchsorted(keys, q)
    ok = in_box(hr[:, 0], hr[:, 1], hr[:, 2]) & (pos < len(keys))
    ok &= keys[np.minimum(pos, len(keys) - 1)] == q
    return np.where(ok, pos, -1).astype(np.int32)
