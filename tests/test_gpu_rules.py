"""-m gpu: rulebook indices must be BIT-EXACT against the oracle (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from helpers import build_metadata, random_points

pytestmark = pytest.mark.gpu


def _cases():
    from occuseg_amd import scenes
    edge = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [1023, 2047, 1023, 0],
                     [1022, 2047, 1023, 0], [1023, 2046, 1022, 0], [5, 5, 5, 0], [5, 5, 5, 0], [5, 5, 5, 0],
                     [0, 0, 0, 1], [1, 1, 1, 1], [1, 1, 1, 1], [2, 2, 2, 1], [3, 3, 3, 1]], np.int64)
    yield "edges+dups", edge
    yield "single-point", np.array([[7, 8, 9]], np.int64)
    yield "dense-8cube-2samples", random_points(0, 400, 2, 8)
    yield "sparse-3samples", random_points(1, 300, 3, 40, origin=10)
    g = scenes.bernoulli_grid(32, 0.25, 0)
    yield "bernoulli32", g
    c, _ = scenes.batch([1000, 1001], target_sites=8000)
    yield "scene-2x8k", c


@pytest.mark.parametrize("path", ["sort_hash", "bitmaps"])
@pytest.mark.parametrize("name,coords", list(_cases()), ids=[n for n, _ in _cases()])
def test_rulebooks_bit_exact(gpu, name, coords, monkeypatch, path):
    """every grid, the input rules and the rulebooks of four scales against the oracle -- built one at a time by sort + hash as
    the ops ask (with the probe table kept), and from occupancy bitmaps (round 6: the first batch on an input size builds
    everything a UNet could ask for ahead, scn_batch_build_occ)"""
    from occuseg_amd import SCN
    from oracle import rules as R
    if path == "sort_hash":
        monkeypatch.setattr(SCN, "KEEP_NBR", True)  # the probe table is dropped after the build unless a test asks
    else:
        monkeypatch.setattr(SCN, "OCC_BUILD", True)
        SCN._PLANS.clear()
    before = dict(SCN.OCC_STATS)
    m, size = build_metadata(coords, gpu)
    if path == "sort_hash":
        assert SCN.OCC_STATS == before
    elif name == "edges+dups":  # sites in the corners of the key box, two samples: beyond SCN_OCC_MAX_WORDS, left to the sort
        assert SCN.OCC_STATS["too_large"] == before["too_large"] + 1 and SCN.OCC_STATS["built"] == before["built"]
    else:
        assert SCN.OCC_STATS["built"] == before["built"] + 1
    grid, header, table, site_of_point = R.input_layer_rules(coords)
    # site order and coordinates
    loc = m.getSpatialLocations(size).numpy()
    assert loc.shape == grid.coords.shape
    assert np.array_equal(loc, grid.coords)
    # input rules (header + rows; padding beyond cnt is unspecified in the reference, 0 in both here)
    h, t = m.getInputRulesHost()
    assert np.array_equal(h, header)
    assert np.array_equal(t, table)
    assert np.array_equal(m.input_site_of_row.cpu().numpy().astype(np.int64), site_of_point)
    f3 = torch.tensor([3, 3, 3])
    f2 = torch.tensor([2, 2, 2])
    for level in range(4):
        rb = m.getSubmanifoldRuleBook(size, f3)
        want, nbr = R.submanifold_rules(grid)
        got = rb.host_lists()
        for k in range(27):
            assert np.array_equal(got[k], want[k]), (name, level, k)
        if path == "sort_hash":
            assert np.array_equal(rb.nbr.cpu().numpy().T, nbr)
        bases = rb.bases[1].cpu().numpy()
        for k in range(27):
            assert np.array_equal(bases[k], np.searchsorted(want[k][:, 1], np.arange(bases.shape[1]) * 32, side="left"))
        if int(size[0]) % 2 or grid.n == 0:
            break
        out_size = size // 2
        rb2 = m.getRuleBook(size, out_size, f2, f2)
        coarse, want2, child = R.strided_rules(grid)
        got2 = rb2.host_lists()
        for k in range(8):
            assert np.array_equal(got2[k], want2[k]), (name, level, k)
        assert m.getNActive(out_size) == coarse.n
        assert np.array_equal(m.getSpatialLocations(out_size).numpy(), coarse.coords)
        # tile-base tables: lower bounds on the output column at granules of 32 rows
        for col, n_out in ((1, coarse.n), (0, grid.n)):
            bases = rb2.bases[col].cpu().numpy()
            for k in range(8):
                ref = np.searchsorted(want2[k][:, col], np.arange(bases.shape[1]) * 32, side="left")
                assert np.array_equal(bases[k], ref)
        grid, size = coarse, out_size


def test_out_of_box_is_an_error(gpu):
    from occuseg_amd import SCN
    m = SCN.Metadata_3()
    bad = torch.tensor([[1024, 0, 0, 0]], dtype=torch.int64)
    with pytest.raises(RuntimeError):
        m.inputLayer(torch.tensor([4096] * 3), bad, 0, 4, gpu)
    neg = torch.tensor([[0, -1, 0, 0]], dtype=torch.int64)
    with pytest.raises(RuntimeError):
        m.inputLayer(torch.tensor([4096] * 3), neg, 0, 4, gpu)


def test_float_and_int32_coordinates(gpu):
    """the collate functions hand over float (or int) coordinates and the reference's InputLayer converts
    them with `.type(torch.LongTensor)` on the host.  Here float32 / int32 device tensors are consumed directly:
    same sites, same order, same input rules as the int64 path on the truncated values; NaN / out-of-box raise."""
    from occuseg_amd import SCN, scenes
    c64, _ = scenes.batch([1000, 1001], target_sites=8000)
    rng = np.random.default_rng(0)
    frac = rng.uniform(0.0, 0.999, size=c64.shape).astype(np.float32)
    frac[:, 3] = 0.0
    cf = c64.astype(np.float32) + frac          # truncation toward zero gives c64 back (all coordinates >= 0)
    assert np.array_equal(cf.astype(np.int64), c64)
    size = torch.tensor([4096] * 3)
    ref = SCN.Metadata_3()
    ref.inputLayer(size, torch.from_numpy(c64).to(gpu), 0, 4, gpu)
    for t in (torch.from_numpy(cf).to(gpu), torch.from_numpy(c64.astype(np.int32)).to(gpu), torch.from_numpy(cf)):
        m = SCN.Metadata_3()
        m.inputLayer(size, t, 0, 4, gpu)
        assert m.getNActive(size) == ref.getNActive(size)
        assert torch.equal(m.getSpatialLocations(size), ref.getSpatialLocations(size))
        assert torch.equal(m.input_order, ref.input_order)
        assert torch.equal(m.input_site_of_row, ref.input_site_of_row)
        assert torch.equal(m.input_site_start, ref.input_site_start)
    for bad in ([[float("nan"), 0, 0, 0]], [[0, 3e9, 0, 0]], [[0, 0, -1.5, 0]], [[1024.5, 0, 0, 0]]):
        with pytest.raises(RuntimeError):
            SCN.Metadata_3().inputLayer(size, torch.tensor(bad, dtype=torch.float32), 0, 4, gpu)
    # -0.5 truncates to 0 (as .long() does): valid
    SCN.Metadata_3().inputLayer(size, torch.tensor([[-0.5, 0.25, 0.75, 0]], dtype=torch.float32), 0, 4, gpu)
    # through the module, float coordinates on the device
    import occuseg_amd
    scn = occuseg_amd.install()
    layer = scn.InputLayer(3, 4096, mode=4)
    feats = torch.randn(c64.shape[0], 3, device=gpu)
    a = layer([torch.from_numpy(cf).to(gpu), feats, None, 2])
    b = layer([torch.from_numpy(c64), feats, None, 2])
    assert torch.equal(a.features, b.features)


def test_rulebooks_against_hand_derived_literal_vectors(gpu):
    """The HIP path against tests/golden/literal_rules.py -- vectors written out by hand from the reference's CUDA
    source (not produced by oracle/rules.py): site order, input rules, 27 + 8 lists, coarse grid, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import literal_rules as L
    m, size = build_metadata(L.POINTS, gpu)
    assert np.array_equal(m.getSpatialLocations(size).numpy(), L.SITE_COORDS)
    h, t = m.getInputRulesHost()
    assert np.array_equal(h, L.INPUT_HEADER) and np.array_equal(t, L.INPUT_TABLE)
    assert np.array_equal(m.input_site_of_row.cpu().numpy().astype(np.int64), L.SITE_OF_POINT)
    got = m.getSubmanifoldRuleBook(size, torch.tensor([3, 3, 3])).host_lists()
    for k, want in enumerate(L.lists(L.SUBMANIFOLD, 27)):
        assert np.array_equal(got[k], want), k
    rb = m.getRuleBook(size, size // 2, torch.tensor([2, 2, 2]), torch.tensor([2, 2, 2]))
    assert np.array_equal(m.getSpatialLocations(size // 2).numpy(), L.COARSE_COORDS)
    for k, want in enumerate(L.lists(L.STRIDED, 8)):
        assert np.array_equal(rb.host_lists()[k], want), k


def test_tile_passes_prefix_and_stream_k_changes_nothing(gpu):
    """scn_tile_passes: prefix sums over the 128-row tiles of max(1, sum_k ceil(pairs_k(tile) / 32)) -- and scn_conv_forward's
    result does not depend on whether the launch draws slices of that sequence (stream-K) or whole tiles (bitwise), and
    the stream-K control block is all zero again after every launch."""
    from occuseg_amd import SCN, _lib
    from occuseg_amd import scenes
    coords, _ = scenes.batch([1000, 1001], 70000)
    size = torch.tensor([4096] * 3)
    f3 = torch.tensor([3, 3, 3])
    outs = []
    before = SCN.STREAM_K
    try:
        for mode in (False, True):
            SCN.STREAM_K = mode
            md = SCN.Metadata_3()
            md.inputLayer(size, torch.from_numpy(coords), 0, 4, gpu)
            rb = md.getSubmanifoldRuleBook(size, f3)
            n = md.getNActive(size)
            n_tiles = (n + 127) // 128
            torch.manual_seed(0)
            x = torch.randn(n, 64, device=gpu)
            w = torch.randn(27, 64, 64, device=gpu) * 0.05
            out = torch.empty(0, device=gpu)
            assert _lib.load().scn_conv_plan(27, n, n, n, 64, 64, 0) == 401
            for _ in range(3):
                SCN.SubmanifoldConvolution_updateOutput(size, f3, md, x, out, w, torch.empty(0, device=gpu), 1)
                assert int(SCN._sk_state(gpu).view(torch.int32).abs().sum().item()) == 0
            outs.append(out)
            # the tables are built when a launch of the big kernel first asks for them (and from then on with the rulebook)
            assert (1 in rb.passes) == mode and 0 not in rb.passes
            if mode:
                assert (27, 1) in SCN._SK_WANTED
                passes = rb.passes[1].cpu().numpy().astype(np.int64)[:n_tiles + 1]
                bases = rb.bases[1].cpu().numpy().astype(np.int64)
                g = np.minimum(np.arange(n_tiles + 1) * 4, bases.shape[1] - 1)
                per_tile = np.maximum(((bases[:, g[1:]] - bases[:, g[:-1]] + 31) // 32).sum(0), 1)
                assert np.array_equal(passes, np.concatenate([[0], np.cumsum(per_tile)]))
                md2 = SCN.Metadata_3()
                md2.inputLayer(size, torch.from_numpy(coords), 0, 4, gpu)
                assert 1 in md2.getSubmanifoldRuleBook(size, f3).passes  # (... with the rulebook)
    finally:
        SCN.STREAM_K = before
    assert torch.equal(outs[0], outs[1])


def _walk_levels(m, size, levels):
    """what a UNet's forward pass asks of the metadata: 3x3x3 rules of every scale, 2/2 rules between them"""
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    out = []
    for level in range(levels):
        rb = m.getSubmanifoldRuleBook(size, f3)
        g = m.grids[tuple(int(v) for v in size)]
        out.append((g.n, g.coords.clone(), g.keys.clone(), rb.host_lists(), rb.counts.clone(), rb.total_pairs, rb.bases[1].clone()))
        if level + 1 < levels:
            rb2 = m.getRuleBook(size, size // 2, f2, f2)
            out.append((rb2.host_lists(), rb2.counts.clone(), rb2.bases[0].clone(), rb2.bases[1].clone()))
            size = size // 2
    out.append((m.input_order.clone(), m.input_site_of_row.clone(), m.input_site_start.clone()))
    return out


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.float32])
def test_grid_chain_one_read_back_same_grids(gpu, monkeypatch, dtype):
    """scn_grid_chain_counts: the second batch on an input size counts the sites of all six scales in one launch and
    builds every grid without a read-back of its own -- the same grids, rulebooks and pair counts as the per-scale
    path, which is the one test_rulebooks_bit_exact holds against the oracle."""
    from occuseg_amd import SCN, scenes
    c, _ = scenes.batch([1000, 1001], target_sites=8000)
    coords = torch.from_numpy(np.ascontiguousarray(c)).to(dtype)
    size = torch.tensor([4096] * 3, dtype=torch.int64)
    lib = SCN._lib_()
    calls = {"scn_grid_build": 0, "scn_grid_build_async": 0, "scn_grid_chain_counts": 0, "scn_batch_build": 0,
             "scn_submanifold_rules": 0, "scn_strided_rules": 0, "scn_tile_bases": 0, "scn_hash_build": 0,
             "scn_occ_bounds": 0, "scn_occ_chain": 0, "scn_batch_build_occ": 0}
    for name in calls:
        real = getattr(lib, name)
        monkeypatch.setattr(lib, name, lambda *a, _r=real, _n=name: (calls.__setitem__(_n, calls[_n] + 1), _r(*a))[1])

    def run(chain, batch=False, occ=False):
        monkeypatch.setattr(SCN, "GRID_CHAIN", chain)
        monkeypatch.setattr(SCN, "BATCH_BUILD", batch)
        monkeypatch.setattr(SCN, "OCC_BUILD", occ)
        for k in calls:
            calls[k] = 0
        m = SCN.Metadata_3()
        m.inputLayer(size, coords, 0, 4, gpu)
        res = _walk_levels(m, size, 6)
        assert not m._grid_status  # every grid built on a counted size has been checked against its own count
        return res, dict(calls)

    SCN._PLANS.pop((4096,) * 3, None)
    ref, n0 = run(False)               # no plan yet: one read-back per scale
    grids = {k: n0[k] for k in ("scn_grid_build", "scn_grid_build_async", "scn_grid_chain_counts", "scn_batch_build")}
    assert grids == {"scn_grid_build": 6, "scn_grid_build_async": 0, "scn_grid_chain_counts": 0, "scn_batch_build": 0}
    assert n0["scn_submanifold_rules"] == 6 and n0["scn_strided_rules"] == 5 and n0["scn_hash_build"] == 6 and n0["scn_tile_bases"] == 16
    again, n1 = run(False)             # plan replayed, chain off
    assert n1 == n0
    got, n2 = run(True)                # plan replayed with the chain
    assert n2 == dict(n0, scn_grid_build=0, scn_grid_build_async=6, scn_grid_chain_counts=1)
    # round 5: ... and every grid and rulebook of the chain queued by ONE call into one arena (scn_batch_build)
    one, n3 = run(True, batch=True)
    assert n3 == dict({k: 0 for k in calls}, scn_grid_chain_counts=1, scn_batch_build=1)
    # round 6: ... from occupancy bitmaps of the batch's bounding box: no sort, no hash map, with the chain switch on or off
    occ, n4 = run(True, batch=True, occ=True)
    assert n4 == dict({k: 0 for k in calls}, scn_occ_bounds=1, scn_occ_chain=1, scn_batch_build_occ=1)
    occ2, n5 = run(False, batch=True, occ="auto")
    assert n5 == n4
    assert [r[0] for r in ref[0::2]] == [r[0] for r in got[0::2]] and ref[0][0] > ref[2][0] > ref[10][0] > 0
    for a, b, d, e, f, g in zip(ref, again, got, one, occ, occ2):
        for x, y, z, u, v, w in zip(a, b, d, e, f, g):
            if torch.is_tensor(x):
                assert torch.equal(x, y) and torch.equal(x, z) and torch.equal(x, u) and torch.equal(x, v) and torch.equal(x, w)
            elif isinstance(x, list):  # the lists of a rulebook (the rows behind counts[k] are unwritten memory)
                assert len(x) == len(y) == len(z) == len(u) == len(v) == len(w)
                assert all(np.array_equal(i, j) and np.array_equal(i, k) and np.array_equal(i, q) and np.array_equal(i, r) and np.array_equal(i, t)
                           for i, j, k, q, r, t in zip(x, y, z, u, v, w))
            else:
                assert x == y == z == u == v == w
    # the error a coordinate outside the key box raises comes from the chain now
    bad = coords.clone()
    bad[5, 1] = 2048
    monkeypatch.setattr(SCN, "GRID_CHAIN", True)
    for batch, occ_ in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(SCN, "BATCH_BUILD", batch)
        monkeypatch.setattr(SCN, "OCC_BUILD", occ_)
        with pytest.raises(RuntimeError, match="key box"):
            SCN.Metadata_3().inputLayer(size, bad, 0, 4, gpu)


def _same_walk(a, b):
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            if torch.is_tensor(u):
                assert torch.equal(u, v)
            elif isinstance(u, list):
                assert len(u) == len(v) and all(np.array_equal(i, j) for i, j in zip(u, v))
            else:
                assert u == v


@pytest.mark.parametrize("case", ["word_edges", "odd_origin_three_samples", "ncols3_float", "long_segment", "sparse_far_apart",
                                  "single_scale", "key_box_corners"])
def test_occupancy_build_against_the_sort_build(gpu, monkeypatch, case):
    """scn_batch_build_occ (round 6: ranks from population counts over bitmaps of the bounding box) against the per-scale sort +
    hash build that test_rulebooks_bit_exact holds against the oracle: every grid, rulebook, tile-base table and the input
    layer's order / site_of_row / site_start, bit for bit -- on clouds made to hit what the bitmaps could get wrong: sites at
    both ends of the 32-voxel words, sites in the corners of the key box (neighbours beyond its faces), a box whose origin is not aligned to the coarsest scale, sample indices that do not start at
    0, three-column and float coordinates, a voxel with more rows than the segment sort takes (the host's repair), and a cloud
    whose box is mostly empty; a plan of one scale (a network without strided convolutions)."""
    from occuseg_amd import SCN
    rng = np.random.default_rng(11)
    levels, dtype, ncols = 4, torch.int64, 4
    if case == "word_edges":
        # dense slabs around x = 31 | 32 and 63 | 64 (relative to an origin of 8 * k): every row has neighbours across a word edge
        x = np.concatenate([np.arange(24, 40), np.arange(56, 72)]) + 160
        g = np.stack(np.meshgrid(x, np.arange(100, 112), np.arange(40, 52), np.arange(2), indexing="ij"), -1).reshape(-1, 4)
        c = g[rng.random(len(g)) < 0.7]
    elif case == "odd_origin_three_samples":
        c = np.concatenate([rng.integers(0, 90, size=(30000, 3)) + np.array([37, 1001, 515]), rng.integers(2, 5, size=(30000, 1))], 1)
    elif case == "ncols3_float":
        c = (rng.integers(0, 60, size=(20000, 3)) + 5).astype(np.float32) + 0.25
        dtype, ncols = torch.float32, 3
    elif case == "long_segment":
        c = np.concatenate([rng.integers(0, 40, size=(8000, 3)) + 3, rng.integers(0, 2, size=(8000, 1))], 1)
        c[rng.permutation(8000)[:300]] = np.array([17, 9, 30, 1])     # 300 rows in one voxel (> SCN_OCC_SEGMENT_MAX = 64)
        c[rng.permutation(8000)[:64]] = np.array([5, 5, 5, 0])        # <= 64: sorted on the device
    elif case == "single_scale":
        c = np.concatenate([rng.integers(0, 70, size=(40000, 3)) + 11, rng.integers(0, 3, size=(40000, 1))], 1)
        levels = 1
    elif case == "key_box_corners":  # the whole key box as the bounding box (one sample: 77 M words), neighbours beyond its faces
        c = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [1023, 2047, 1023, 0], [1022, 2047, 1023, 0],
                      [1023, 2046, 1022, 0], [1022, 2046, 1022, 0], [5, 5, 5, 0], [5, 5, 5, 0], [31, 5, 5, 0], [32, 5, 5, 0]], np.int64)
    else:
        c = np.concatenate([rng.integers(0, 1000, size=(3000, 3)), rng.integers(0, 2, size=(3000, 1))], 1)
    coords = torch.from_numpy(np.ascontiguousarray(c)).to(dtype)
    size = torch.tensor([4096 + 16 * (1 + ["word_edges", "odd_origin_three_samples", "ncols3_float", "long_segment",
                                            "sparse_far_apart", "single_scale", "key_box_corners"].index(case))] * 3, dtype=torch.int64)
    monkeypatch.setattr(SCN, "GRID_CHAIN", False)

    def run(occ):
        monkeypatch.setattr(SCN, "OCC_BUILD", occ)
        m = SCN.Metadata_3()
        m.inputLayer(size, coords, 0, 4, gpu)
        return _walk_levels(m, size, levels), m

    SCN._PLANS.pop(tuple(int(v) for v in size), None)
    ref, _ = run(False)          # no plan yet: per-scale builds
    ref2, _ = run(False)         # plan replayed, per-scale builds
    before = dict(SCN.OCC_STATS)
    got, m = run(True)
    assert SCN.OCC_STATS["built"] == before["built"] + 1
    assert SCN.OCC_STATS["order_repaired"] == before["order_repaired"] + (1 if case == "long_segment" else 0)
    _same_walk(ref, ref2)
    _same_walk(ref, got)
    # a grid built from bitmaps has no hash map until something asks for one
    g0 = m.grids[tuple(int(v) for v in size)]
    assert g0.hash_keys is None and m._hash(g0).hash_keys is not None
    auto, _ = run("auto")
    _same_walk(ref, auto)
    if case in ("sparse_far_apart", "key_box_corners"):  # 3,000 points in a 1000^3 box (12 in the key box): "auto" leaves it to the sort
        assert SCN.OCC_STATS["too_large"] == before["too_large"] + 1 and SCN.OCC_STATS["built"] == before["built"] + 1
    else:
        assert SCN.OCC_STATS["built"] == before["built"] + 2


def test_occupancy_chain_abi(gpu):
    """scn_occ_bounds / scn_occ_plan / scn_occ_chain by themselves: bounds and site counts of coords >> l against numpy, the
    high halves of the words = running population counts, duplicates, sample indices from 3, 1 to 12 scales"""
    import ctypes
    from occuseg_amd import SCN, _lib
    from occuseg_amd.SCN import ptr, stream, check
    lib = SCN._lib_()
    rng = np.random.default_rng(6)
    for P, span, ncols, levels, dt in ((1, 8, 3, 1, np.int32), (5000, 40, 4, 12, np.int64), (200000, 300, 4, 6, np.int32)):
        c = (rng.integers(0, span, size=(P, ncols)) + 7).astype(dt)
        if ncols == 4:
            c[:, 3] = rng.integers(3, 6, size=P)
        t = torch.from_numpy(c).to(gpu)
        mm = (ctypes.c_int32 * 8)()
        scratch = torch.empty(16, dtype=torch.int32, device=gpu)
        check(lib.scn_occ_bounds(ptr(t), 0 if dt == np.int32 else 1, ncols, P, ptr(scratch), mm, stream()))
        want = list(c[:, :3].min(0)) + [c[:, 3].min() if ncols == 4 else 0] + list(c[:, :3].max(0)) + [c[:, 3].max() if ncols == 4 else 0]
        assert list(mm) == [int(v) for v in want]
        box = _lib.OccBox()
        check(lib.scn_occ_plan(mm, levels, ctypes.byref(box)))
        assert box.bytes > 0 and all(box.origin[a] % (1 << (levels - 1)) == 0 and box.origin[a] <= mm[a] for a in range(3))
        buf = torch.zeros(box.bytes, dtype=torch.uint8, device=gpu)
        n = (ctypes.c_int32 * levels)()
        check(lib.scn_occ_chain(ptr(t), 0 if dt == np.int32 else 1, ncols, P, ctypes.byref(box), ptr(buf), box.bytes, n, stream()))
        for l in range(levels):
            cc = c.copy()
            cc[:, :3] >>= l
            assert n[l] == len(np.unique(cc, axis=0)), (P, l)
        W = int(box.woff[levels])
        words = buf[box.off_words:box.off_words + 8 * W].view(torch.int64).cpu().numpy().astype(np.uint64)
        bits = (words & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        pop = np.array([bin(int(b)).count("1") for b in bits], np.int64)
        assert np.array_equal((words >> np.uint64(32)).astype(np.int64), np.concatenate([[0], np.cumsum(pop)[:-1]]))
        assert pop.sum() == sum(n)
    # a box that does not fit: bytes = 0, no error
    mm = (ctypes.c_int32 * 8)(0, 0, 0, 0, 1023, 2047, 1023, 63)
    box = _lib.OccBox()
    check(lib.scn_occ_plan(mm, 6, ctypes.byref(box)))
    assert box.bytes == 0


def test_grid_chain_counts_abi(gpu):
    """the C entry by itself: counts of coords >> l against numpy's unique, duplicates, three samples, 12 scales"""
    import ctypes
    from occuseg_amd import SCN
    from occuseg_amd.SCN import ptr, stream, check
    lib = SCN._lib_()
    rng = np.random.default_rng(5)
    for P, span, ncols in ((1, 8, 3), (5000, 40, 4), (200000, 900, 4)):
        c = rng.integers(0, span, size=(P, ncols)).astype(np.int32)
        if ncols == 4:
            c[:, 3] = rng.integers(0, 3, size=P)
        levels = 12
        t = torch.from_numpy(c).to(gpu)
        tb = lib.scn_grid_chain_temp_bytes(P, levels)
        temp = torch.empty(tb, dtype=torch.uint8, device=gpu)
        n = (ctypes.c_int32 * levels)()
        check(lib.scn_grid_chain_counts(ptr(t), 0, ncols, P, levels, ptr(temp), tb, n, stream()))
        for l in range(levels):
            cc = c.copy()
            cc[:, :3] >>= l
            assert n[l] == len(np.unique(cc, axis=0)), (P, l)
    assert lib.scn_grid_chain_temp_bytes(100, 13) == 0 and lib.scn_grid_chain_temp_bytes(0, 2) == 0


def test_batch_layout_query(gpu):
    """scn_batch_layout_query (ABI v6): the arena of a batch's grids and rulebooks -- offsets 256-byte aligned, disjoint, in order,
    results in front of the scratch; bad plans are refused.  (No launch, but rocPRIM's workspace query wants a device.)"""
    import ctypes
    from occuseg_amd import _lib
    lib = _lib.load()
    plan = _lib.BatchPlan()
    plan.levels, plan.coords_dtype, plan.ncols, plan.P = 3, 1, 4, 100000
    for l, n in enumerate((60000, 17000, 4000)):
        plan.N[l], plan.sub[l], plan.strided[l] = n, 1, 1 if l < 2 else 0
    lo = _lib.BatchLayout()
    assert lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)) == 0
    spans = [(lo.small, 3 * 37 * 4), (lo.order, 4 * 100000), (lo.site_of_row, 4 * 100000), (lo.site_start, 4 * 100001)]
    for l, n in enumerate((60000, 17000, 4000)):
        g = (n + 31) // 32
        cap = int(lo.hash_cap[l])
        assert cap == lib.scn_hash_capacity(n)
        spans += [(lo.coords[l], 16 * n), (lo.keys[l], 8 * n), (lo.hash_keys[l], 8 * cap), (lo.hash_vals[l], 4 * cap),
                  (lo.sub_rules[l], 27 * n * 8), (lo.sub_bases[l], 27 * (g + 1) * 4)]
        if l < 2:
            gc = ((17000, 4000)[l] + 31) // 32
            spans += [(lo.str_rules[l], 8 * n * 8), (lo.str_bases_fine[l], 8 * (g + 1) * 4), (lo.str_bases_coarse[l], 8 * (gc + 1) * 4)]
        else:
            assert lo.str_rules[l] == lo.str_bases_fine[l] == lo.str_bases_coarse[l] == -1
    spans.sort()
    for (a, na), (b, _) in zip(spans, spans[1:]):
        assert a % 256 == 0 and a >= 0 and a + na <= b, (a, na, b)
    assert spans[-1][0] + spans[-1][1] <= lo.scratch <= lo.total and lo.scratch % 256 == 0
    # the scratch holds at least the level-0 grid build's workspace and the largest neighbour table
    assert lo.total - lo.scratch >= lib.scn_grid_build_temp_bytes(100000) + 8 * 100000 + 27 * 60000 * 4
    for l in range(3, _lib.BATCH_MAX_LEVELS):
        assert lo.coords[l] == -1 and lo.hash_cap[l] == 0
    # refused: no levels, site counts that grow, a strided rulebook without a next scale
    plan.levels = 0
    assert lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)) != 0
    plan.levels = 3
    plan.N[1] = 70000
    assert lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)) != 0 and b"non-increasing" in lib.scn_last_error()
    plan.N[1] = 17000
    plan.strided[2] = 1
    assert lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)) != 0


def test_prefetched_bitmap_build_repairs_a_long_segment_at_adoption(gpu, monkeypatch):
    """a batch built ahead on the side stream under no_grad leaves its read-back to the InputLayer call that adopts it
    (SCN.DEFER_PREFETCH_READBACK): the "a voxel holds more than 64 rows" flag of scn_batch_build_occ arrives there too, and the
    host's sort of (site, row) runs before anything reads `order` -- the adopted arrays equal the sort build's"""
    from occuseg_amd import SCN
    rng = np.random.default_rng(3)
    c = np.concatenate([rng.integers(0, 50, size=(12000, 3)) + 4, rng.integers(0, 2, size=(12000, 1))], 1)
    c[rng.permutation(12000)[:200]] = np.array([20, 21, 22, 1])
    coords = torch.from_numpy(np.ascontiguousarray(c)).to(gpu)
    size = torch.tensor([4096 + 16 * 11] * 3, dtype=torch.int64)
    key = tuple(int(v) for v in size)
    monkeypatch.setattr(SCN, "GRID_CHAIN", False)
    monkeypatch.setattr(SCN, "OCC_BUILD", False)
    SCN._PLANS.pop(key, None)
    m0 = SCN.Metadata_3()
    m0.inputLayer(size, coords, 0, 4, gpu)
    ref = _walk_levels(m0, size, 3)          # (records the plan)
    monkeypatch.setattr(SCN, "OCC_BUILD", True)
    before = dict(SCN.OCC_STATS)
    with torch.no_grad():
        SCN.prefetch_input(size, coords, 4, gpu)
        pre = SCN._PREFETCHED[id(coords)][2]
        assert pre._deferred is not None and SCN.OCC_STATS["order_repaired"] == before["order_repaired"]  # (not read back yet)
        m = SCN.Metadata_3()
        m.inputLayer(size, coords, 0, 4, gpu)
    assert SCN.OCC_STATS["built"] == before["built"] + 1 and SCN.OCC_STATS["order_repaired"] == before["order_repaired"] + 1
    _same_walk(ref, _walk_levels(m, size, 3))


def test_batch_build_occ_with_and_without_a_scratch_of_its_own(gpu):
    """scn_batch_build_occ's two forms -- results and scratch in one arena of layout.total bytes (scratch = NULL), or the scratch as an
    allocation the caller drops behind the call -- leave the same results, byte for byte"""
    import ctypes
    from occuseg_amd import SCN, _lib, scenes
    from occuseg_amd.SCN import ptr, stream, check
    lib = SCN._lib_()
    c, _ = scenes.batch([1000, 1001], target_sites=8000)
    t = torch.from_numpy(np.ascontiguousarray(c)).to(gpu)
    P, L = t.size(0), 3
    mm = (ctypes.c_int32 * 8)()
    check(lib.scn_occ_bounds(ptr(t), 1, 4, P, ptr(torch.empty(16, dtype=torch.int32, device=gpu)), mm, stream()))
    box = _lib.OccBox()
    check(lib.scn_occ_plan(mm, L, ctypes.byref(box)))
    results = []
    for own_scratch in (False, True):
        buf = torch.empty(box.bytes, dtype=torch.uint8, device=gpu)
        n = (ctypes.c_int32 * L)()
        check(lib.scn_occ_chain(ptr(t), 1, 4, P, ctypes.byref(box), ptr(buf), box.bytes, n, stream()))
        plan = _lib.BatchPlan()
        plan.levels, plan.coords_dtype, plan.ncols, plan.P, plan.flags = L, 1, 4, P, 1
        for l in range(L):
            plan.N[l], plan.sub[l], plan.strided[l] = n[l], 1, 1 if l + 1 < L else 0
        lo = _lib.BatchLayout()
        check(lib.scn_batch_layout_query(ctypes.byref(plan), ctypes.byref(lo)))
        if own_scratch:
            arena = torch.full((lo.scratch,), 0xA5, dtype=torch.uint8, device=gpu)
            scratch = torch.empty(lo.total - lo.scratch, dtype=torch.uint8, device=gpu)
            check(lib.scn_batch_build_occ(ptr(t), ctypes.byref(plan), ctypes.byref(box), ptr(buf), ptr(arena), lo.scratch, ptr(scratch),
                                          lo.total - lo.scratch, stream()))
            # a scratch that is too small is refused
            assert lib.scn_batch_build_occ(ptr(t), ctypes.byref(plan), ctypes.byref(box), ptr(buf), ptr(arena), lo.scratch, ptr(scratch),
                                           lo.total - lo.scratch - 1, stream()) != 0
        else:
            arena = torch.full((lo.total,), 0xA5, dtype=torch.uint8, device=gpu)
            check(lib.scn_batch_build_occ(ptr(t), ctypes.byref(plan), ctypes.byref(box), ptr(buf), ptr(arena), lo.total, None, 0, stream()))
            assert lib.scn_batch_build_occ(ptr(t), ctypes.byref(plan), ctypes.byref(box), ptr(buf), ptr(arena), lo.scratch, None, 0,
                                           stream()) != 0
        results.append((arena[:lo.scratch].cpu(), [int(v) for v in n]))
    (a, na), (b, nb) = results
    assert na == nb
    # every byte the build writes is the same (rows behind a list's length and alignment gaps keep the 0xA5 fill in both)
    assert torch.equal(a, b)
