"""not gpu: the oracle itself -- two independent rule generators agree, structural invariants of
SURVEY.md Appendix A hold, arithmetic matches the reference's own CPU loops (oracle/_ref) and the
committed golden vectors, hand-derived backward formulas match torch autograd."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, random_points

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    from occuseg_amd import scenes
    yield "dense", random_points(0, 300, 2, 7)
    yield "sparse", random_points(1, 200, 3, 30, origin=10)
    yield "box-edges", np.array([[0, 0, 0, 0], [1023, 2047, 1023, 0], [1022, 2046, 1022, 0], [1023, 2047, 1022, 0],
                                 [0, 2047, 0, 1], [1, 2047, 0, 1], [1023, 0, 0, 1]], np.int64)
    yield "bernoulli", np.concatenate([scenes.bernoulli_grid(12, 0.3, 3), np.zeros((0, 3), np.int64)])


@pytest.mark.parametrize("name,coords", list(_cases()), ids=[n for n, _ in _cases()])
def test_rules_numpy_vs_bruteforce(name, coords):
    from oracle import bruteforce as BF, rules as R
    grid, header, table, sop = R.input_layer_rules(coords)
    c4 = [tuple(int(v) for v in (list(r) + [0])[:4]) for r in coords.tolist()]
    sc, ctr, rows, ma = BF.sites_and_input_rules(c4)
    assert [tuple(r) for r in grid.coords.tolist()] == sc
    assert list(grid.ctr) == ctr and header.tolist() == [4, ma, len(coords), len(sc)]
    for s, r in enumerate(rows):
        assert table[s, 0] == r[0] and table[s, 1:1 + r[0]].tolist() == r[1:]
        assert all(sop[p] == s for p in r[1:])
    sub, nbr = R.submanifold_rules(grid)
    bsub = BF.submanifold_rules(sc)
    for k in range(27):
        assert [tuple(x) for x in sub[k].tolist()] == bsub[k]
    coarse, strd, child = R.strided_rules(grid)
    cc, cctr, bstr = BF.strided_rules(sc)
    assert [tuple(r) for r in coarse.coords.tolist()] == cc and list(coarse.ctr) == cctr
    for k in range(8):
        assert [tuple(x) for x in strd[k].tolist()] == bstr[k]


def test_rule_invariants():
    """lists ascending in `out`; list 13 is the identity; (i,o) in k <=> (o,i) in 26-k;
    every fine site in exactly one strided list; strided lists ascending in BOTH columns."""
    from oracle import rules as R
    coords = random_points(5, 1500, 3, 14)
    grid, _, _, _ = R.input_layer_rules(coords)
    sub, nbr = R.submanifold_rules(grid)
    assert np.array_equal(sub[13], np.stack([np.arange(grid.n)] * 2, 1))
    for k in range(27):
        assert np.all(np.diff(sub[k][:, 1]) > 0)
        a = set(map(tuple, sub[k].tolist()))
        b = set((o, i) for i, o in sub[26 - k].tolist())
        assert a == b
        # no pair crosses a sample boundary
        assert np.array_equal(grid.coords[sub[k][:, 0], 3], grid.coords[sub[k][:, 1], 3])
    coarse, strd, child = R.strided_rules(grid)
    seen = np.concatenate([r[:, 0] for r in strd])
    assert np.array_equal(np.sort(seen), np.arange(grid.n))
    for r in strd:
        assert np.all(np.diff(r[:, 0]) > 0) and np.all(np.diff(r[:, 1]) > 0)
    assert np.array_equal(coarse.coords[:, :3][np.concatenate([r[:, 1] for r in strd])],
                          grid.coords[:, :3][seen] >> 1)


def test_key_order_is_z_y_x():
    from oracle import rules as R
    c = np.array([[5, 0, 0], [0, 1, 0], [0, 0, 1], [1023, 2047, 0], [0, 0, 0]], np.int64)
    grid, _, _, _ = R.input_layer_rules(c)
    assert grid.coords[:, :3].tolist() == [[0, 0, 0], [5, 0, 0], [0, 1, 0], [1023, 2047, 0], [0, 0, 1]]


def test_golden_rulebooks():
    from oracle import rules as R
    for name in ("g1_rules_8cube", "g2_rules_bernoulli32", "g3_rules_bernoulli14"):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        grid, header, table, sop = R.input_layer_rules(g["coords"])
        assert np.array_equal(grid.coords, g["site_coords"]) and np.array_equal(header, g["header"])
        assert np.array_equal(table, g["table"]) and np.array_equal(sop, g["site_of_point"])
        sub, _ = R.submanifold_rules(grid)
        coarse, strd, _ = R.strided_rules(grid)
        assert all(np.array_equal(sub[k], g["sub%02d" % k]) for k in range(27))
        assert all(np.array_equal(strd[k], g["str%d" % k]) for k in range(8))
        assert np.array_equal(coarse.coords, g["coarse_coords"])


def _g3():
    from oracle import rules as R
    g = np.load(os.path.join(GOLD, "g3_conv_bernoulli14.npz"))
    r = np.load(os.path.join(GOLD, "g3_rules_bernoulli14.npz"))
    sub = [r["sub%02d" % k] for k in range(27)]
    strd = [r["str%d" % k] for k in range(8)]
    return g, sub, strd, len(r["site_coords"]), len(r["coarse_coords"])


def test_golden_conv_arithmetic():
    """oracle.arith (float32 and float64) against vectors produced by the reference's own CPU code"""
    from oracle import arith as A
    g, sub, strd, n, nc = _g3()
    for dt, rel in ((np.float32, 2e-6), (np.float64, 2e-6)):
        y, _ = A.conv_forward(g["x"], g["w"], sub, n, dtype=dt)
        assert_close(y, g["y"], "SC fwd", rel)
        dx, dw, _ = A.conv_backward(g["x"], g["dy"], g["w"], sub, dtype=dt)
        assert_close(dx, g["dx"], "SC dInput", rel)
        assert_close(dw, g["dw"], "SC dWeight", rel)
        yc, _ = A.conv_forward(g["x"], g["wc"], strd, nc, dtype=dt)
        assert_close(yc, g["yc"], "conv fwd", rel)
        dxc, dwc, _ = A.conv_backward(g["x"], g["dyc"], g["wc"], strd, dtype=dt)
        assert_close(dxc, g["dxc"], "conv dInput", rel)
        assert_close(dwc, g["dwc"], "conv dWeight", rel)
        yd, _ = A.conv_forward(g["xd"], g["wd"], strd, n, in_col=1, out_col=0, dtype=dt)
        assert_close(yd, g["yd"], "deconv fwd", rel)
        dxd, dwd, _ = A.conv_backward(g["xd"], g["dyd"], g["wd"], strd, in_col=1, out_col=0, dtype=dt)
        assert_close(dxd, g["dxd"], "deconv dInput", rel)
        assert_close(dwd, g["dwd"], "deconv dWeight", rel)


def test_golden_batchnorm_and_io():
    from oracle import arith as A
    g = np.load(os.path.join(GOLD, "g4_batchnorm.npz"))
    for nn in (1024, 2):
        x, gm, b, rm0, rv0, dy = (g["%s%d" % (k, nn)] for k in ("x", "g", "b", "rm", "rv", "dy"))
        for li, leak in enumerate((0.0, 0.333, 1.0)):
            for train in (1, 0):
                tag = "%d_l%d_t%d" % (nn, li, train)
                y, sm, si, rm, rv = A.batchnorm_forward(x, gm, b, rm0, rv0, 1e-4, 0.9, bool(train), leak, np.float64)
                # the reference sums x^2 in float32 (BatchNormalization.cpp:26): allow its own rounding
                assert_close(y, g["y" + tag], "bn y " + tag, 2e-5)
                assert_close(sm, g["sm" + tag], "bn mean", 2e-5)
                assert_close(si, g["si" + tag], "bn invstd", 2e-5)
                assert_close(rm, g["rmo" + tag], "bn running_mean", 2e-5)
                assert_close(rv, g["rvo" + tag], "bn running_var", 2e-5)
                if train:
                    dx, dm, dw, db = A.batchnorm_backward(x, g["y" + tag], dy, g["sm" + tag], g["si" + tag], gm, leak,
                                                          np.float64)
                    scale = max(float(np.abs(dm * g["si" + tag] * gm).max()), float(np.abs(g["dx" + tag]).max()))
                    assert_close(dx, g["dx" + tag], "bn dInput", 5e-5, scale=scale)
                    assert_close(dm, g["dm" + tag], "bn masked dOut", 1e-6)
                    assert_close(dw, g["dw" + tag], "bn dWeight", 5e-5)
                    assert_close(db, g["db" + tag], "bn dBias", 5e-5)
    io = np.load(os.path.join(GOLD, "g5_io_layers.npz"))
    for mode in (3, 4):
        o = A.input_layer_forward(io["feats"], io["table"], mode, np.float64)
        assert_close(o, io["out_m%d" % mode], "input fwd", 1e-6)
        di = A.input_layer_backward(io["out_m%d" % mode], io["table"], len(io["feats"]), mode, np.float64)
        assert_close(di, io["din_m%d" % mode], "input bwd", 1e-6)


def test_oracle_vs_reference_build_live():
    """when oracle/_ref exists (build container, or shipped to the GPU box) re-check on fresh random data"""
    from oracle import arith as A, ref, rules as R
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference): golden vectors cover this")
    coords = random_points(11, 900, 2, 11)
    grid, _, table, _ = R.input_layer_rules(coords)
    sub, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(3)
    x = rng.normal(size=(grid.n, 24)).astype(np.float32)
    w = rng.normal(size=(27, 24, 8)).astype(np.float32) * 0.1
    dy = rng.normal(size=(grid.n, 8)).astype(np.float32)
    assert_close(A.conv_forward(x, w, sub, grid.n)[0], ref.conv_forward(x, w, sub, grid.n), "fwd", 2e-6)
    dx, dw, _ = A.conv_backward(x, dy, w, sub)
    rdx, rdw = ref.conv_backward(x, dy, w, sub)
    assert_close(dx, rdx, "dInput", 2e-6)
    assert_close(dw, rdw, "dWeight", 2e-6)
    xb = rng.normal(size=(777, 24)).astype(np.float32) + 3
    y, sm, si, rm, rv = A.batchnorm_forward(xb, None, None, np.zeros(24), np.ones(24), leakiness=0.2)
    ry, rsm, rsi, rrm, rrv = ref.bn_forward(xb, None, None, np.zeros(24), np.ones(24), leakiness=0.2)
    assert_close(y, ry, "bn y", 2e-5)
    assert_close(rv, rrv, "bn running var", 2e-5)
    f = rng.normal(size=(len(coords), 5)).astype(np.float32)
    assert_close(A.input_layer_forward(f, table, 4), ref.input_forward(f, table, 4), "input", 1e-6)


def test_backward_formulas_match_autograd():
    """oracle.arith.conv_backward / batchnorm_backward vs torch autograd of the forward formulas"""
    from oracle import arith as A, rules as R
    coords = random_points(7, 500, 2, 9)
    grid, _, _, _ = R.input_layer_rules(coords)
    sub, _ = R.submanifold_rules(grid)
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.normal(size=(grid.n, 6)), requires_grad=True)
    w = torch.tensor(rng.normal(size=(27, 6, 5)), requires_grad=True)
    out = torch.zeros(grid.n, 5, dtype=torch.float64)
    for k, r in enumerate(sub):
        out = out.index_add(0, torch.from_numpy(r[:, 1].astype(np.int64)),
                            x.index_select(0, torch.from_numpy(r[:, 0].astype(np.int64))) @ w[k])
    dy = torch.tensor(rng.normal(size=(grid.n, 5)))
    (out * dy).sum().backward()
    dx, dw, db = A.conv_backward(x.detach().numpy(), dy.numpy(), w.detach().numpy(), sub, want_bias=True,
                                 dtype=np.float64)
    assert_close(dx, x.grad.numpy(), "dInput", 1e-12)
    assert_close(dw, w.grad.numpy(), "dWeight", 1e-12)
    assert_close(db, dy.sum(0).numpy(), "dBias", 1e-12)
    xb = torch.tensor(rng.normal(size=(300, 4)) + 1.0, requires_grad=True)
    g = torch.tensor(rng.uniform(0.5, 2, 4), requires_grad=True)
    b = torch.tensor(rng.normal(size=4), requires_grad=True)
    mean = xb.mean(0)
    var = ((xb - mean) ** 2).mean(0)
    y = (xb - mean) / torch.sqrt(var + 1e-4) * g + b
    y = torch.where(y > 0, y, 0.3 * y)
    dyb = torch.tensor(rng.normal(size=(300, 4)))
    (y * dyb).sum().backward()
    yo, sm, si, _, _ = A.batchnorm_forward(xb.detach().numpy(), g.detach().numpy(), b.detach().numpy(), np.zeros(4),
                                           np.ones(4), 1e-4, 0.9, True, 0.3, np.float64)
    assert_close(yo, y.detach().numpy(), "bn fwd", 1e-10)
    dxo, _, dgo, dbo = A.batchnorm_backward(xb.detach().numpy(), yo, dyb.numpy(), sm, si, g.detach().numpy(), 0.3,
                                            np.float64)
    assert_close(dxo, xb.grad.numpy(), "bn dInput", 1e-9)
    assert_close(dgo, g.grad.numpy(), "bn dWeight", 1e-9)
    assert_close(dbo, b.grad.numpy(), "bn dBias", 1e-9)


def test_oracle_against_hand_derived_literal_vectors():
    """oracle/rules.py against tests/golden/literal_rules.py: site order, input rules, 27 submanifold lists, coarse
    grid and 8 strided lists written out BY HAND from the reference's CUDA source for a 9-site, 2-sample case that
    includes coordinate 0 and the far corner of the key box (the aliasing pair).  Independent of both restatements."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import literal_rules as L
    from oracle import bruteforce as BF, rules as R
    grid, header, table, site_of_point = R.input_layer_rules(L.POINTS)
    assert np.array_equal(grid.coords, L.SITE_COORDS)
    assert np.array_equal(header, L.INPUT_HEADER) and np.array_equal(table, L.INPUT_TABLE)
    assert np.array_equal(site_of_point, L.SITE_OF_POINT)
    rules, _ = R.submanifold_rules(grid)
    for k, (got, want) in enumerate(zip(rules, L.lists(L.SUBMANIFOLD, 27))):
        assert np.array_equal(got, want), k
    coarse, srules, _ = R.strided_rules(grid)
    assert np.array_equal(coarse.coords, L.COARSE_COORDS)
    for k, (got, want) in enumerate(zip(srules, L.lists(L.STRIDED, 8))):
        assert np.array_equal(got, want), k
    # the pairs the reference would add through its unchecked neighbour key: 19 + 15 + 15, all with in = site 6
    extra = L.reference_alias_pairs()
    assert sum(len(v) for v in extra.values()) == 49 and all(i == 6 for v in extra.values() for i, _ in v)
    assert 13 not in extra and 26 not in extra and 22 not in extra


def test_normal_guided_remap_restatement():
    """oracle/rules.py's restatement of remap_rules_with_normal / OrientedFilter / the per-site normal average:
    structural facts the reference code implies (SubmanifoldConvolutionRules.h:213-245, RectangularRegions.h:12-32)."""
    from oracle import rules as R
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.integers(0, 12, (500, 3)), np.zeros((500, 1), np.int64)], 1)
    grid, _, table, _ = R.input_layer_rules(pts)
    pn = rng.normal(size=(500, 3)).astype(np.float32)
    sn = R.site_normals(pn, table)
    assert np.allclose(np.linalg.norm(sn, axis=1), 1.0, atol=1e-5)
    ori = R.oriented_filter(sn)
    assert set(np.unique(ori)) <= {0, 2, 4}
    assert R.oriented_filter(np.array([[1, 1, 1], [0, 2, -2], [0, 0, 0], [-3, 1, 1]], np.float32)).tolist() == [0, 2, 0, 0]
    for o in range(6):  # every class permutes the offsets and keeps the centre
        assert sorted(R.ROT_INDEX[o]) == list(range(27)) and R.ROT_INDEX[o][13] == 13
    rules, _ = R.submanifold_rules(grid)
    new = R.remap_rules_with_normal(rules, ori)
    assert sum(len(r) for r in new) == sum(len(r) for r in rules)
    assert np.array_equal(new[13], rules[13])
    for n, r in enumerate(new):
        assert len(np.unique(r[:, 1])) == len(r)  # an output site at most once per list
        # pairs of class-0 outputs did not move
        keep = rules[n][ori[rules[n][:, 1]] == 0]
        got = r[ori[r[:, 1]] == 0]
        assert np.array_equal(got, keep)
    # all-x normals: nothing moves at all
    same = R.remap_rules_with_normal(rules, np.zeros(grid.n, np.int64))
    assert all(np.array_equal(a, b) for a, b in zip(same, rules))
    # cross-scale lookup
    coarse, _, _ = R.strided_rules(grid)
    idx = R.resolution_based_scattering(coarse.coords[:, :3], grid.coords[:, :3], 2)
    assert np.array_equal(coarse.coords[idx, :3], grid.coords[:, :3] >> 1)


def test_scatter_oracle_formulas():
    from oracle import scatter as O
    src = np.array([[1.0, 2.0], [3.0, 6.0], [5.0, 1.0], [7.0, 7.0]])
    idx = np.array([0, 0, 2, 0])
    assert O.scatter_add(src, idx, 4).tolist() == [[11, 15], [0, 0], [5, 1], [0, 0]]
    assert np.allclose(O.scatter_mean(src, idx, 4), [[11 / 3, 5], [0, 0], [5, 1], [0, 0]])
    assert np.allclose(O.scatter_std(src, idx, 4)[0], np.std(src[[0, 1, 3]], axis=0, ddof=1))
    assert np.allclose(O.scatter_std(src, idx, 4)[2], 0)  # a single element: count - 1 clamped to 1
    mx, arg = O.scatter_max(src, idx, 4)
    assert mx.tolist() == [[7, 7], [0, 0], [5, 1], [0, 0]] and arg.tolist() == [[3, 3], [4, 4], [2, 2], [4, 4]]


def test_reference_loops_network_matches_the_autograd_oracle():
    """bench.py's cpu_baseline ("reference"): oracle/ref_net.py walks the network over the reference's compiled CPU loops
    (oracle/_ref) with an explicit backward pass.  Pinned here against the torch-autograd oracle (float64) on a small
    UNet: it IS the same computation -- forward at 1e-4; gradients at 1e-2 with ReLU (float32 masks flip), at 1e-4
    without."""
    from oracle import ref as REF, ref_net, torch_ref
    if not REF.available():
        pytest.skip("oracle/_ref/libscn_ref.so not built (needs /root/reference)")
    import occuseg_amd
    from occuseg_amd import scenes
    scn = occuseg_amd.install()
    coords, feats = scenes.batch([1000, 1001], 3000)
    for leak, rel in ((0.0, 1e-2), (1.0, 1e-4)):
        torch.manual_seed(3)
        m = 16
        net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
            scn.UNet(3, 1, [m, 2 * m, 3 * m], residual_blocks=True, leakiness=leak)).add(
            scn.BatchNormalization(m, leakiness=leak)).add(scn.OutputLayer(3))
        want, stats = torch_ref.ref_forward(net, coords, feats, torch.float64)
        want.sum().backward()
        got, grads, sec, _ctx = ref_net.forward_backward(net, coords, feats, threads=2)
        assert_close(got, want.detach().numpy(), "reference-loops forward")
        assert sec["total"] > 0 and sec["conv"] > 0
        n = 0
        for mod in net.modules():
            for name, leaf in (stats["params"].get(id(mod)) or {}).items():
                assert_close(grads[(id(mod), name)], leaf.grad.numpy(), "reference-loops grad %s.%s" % (mod, name), rel=rel)
                n += 1
        assert n == len(list(net.parameters()))


def test_bench_parity_fullsize_machinery_on_cpu():
    """bench.py's parity_fullsize (compare_passes + the masked backward pass of oracle/ref_net.py), exercised without a GPU:
    the 'device pass' is the float32 torch-autograd oracle -- another float32 evaluation with another summation order,
    which is what the HIP path is to the reference's loops.  With the device's activation masks handed to the reference's
    BatchNormalization_BackwardPass every gradient holds the 1e-4 bar; a corrupted gradient must be caught and named."""
    from oracle import ref as REF, ref_net, torch_ref
    if not REF.available():
        pytest.skip("oracle/_ref/libscn_ref.so not built (needs /root/reference)")
    import bench
    import occuseg_amd
    from occuseg_amd import scenes
    scn = occuseg_amd.install()
    coords, feats = scenes.batch([1000, 1001], 3000)
    torch.manual_seed(5)
    net = bench.build_net(scn, 16)
    out, stats = torch_ref.ref_forward(net, coords, feats, torch.float32)
    out.sum().backward()
    bns = [mod for mod in net.modules() if isinstance(mod, scn.BatchNormalization)]
    dev = {"out": out.detach().numpy(), "masks": [stats["own_masks"][id(mod)].numpy() for mod in bns], "grads": []}
    for mod in net.modules():
        for name, p in mod._parameters.items():
            if p is not None:
                dev["grads"].append(("%s.%s" % (type(mod).__name__, name), tuple(p.shape),
                                     stats["params"][id(mod)][name].grad.numpy().astype(np.float32)))
    masks = {id(mod): mk for mod, mk in zip(bns, dev["masks"])}
    got, grads, _sec, ctx = ref_net.forward_backward(net, coords, feats, threads=2, masks=masks)
    order = [("%s.%s" % (type(mod).__name__, name), grads[(id(mod), name)])
             for mod in net.modules() for name, p in mod._parameters.items() if p is not None]
    res = bench.compare_passes(dev, got, order, ctx.mask_flips, ctx.mask_total)
    assert res["ok"], res
    assert res["forward_max_rel"] < 1e-4 and res["grad_max_rel"] < 1e-4 and res["grad_tensors"] == len(list(net.parameters()))
    assert res["mask_elements"] == sum(m.size for m in dev["masks"]) and res["mask_flips"] < 1e-3 * res["mask_elements"]
    assert res["elementwise_rel_p99"] is not None and res["elementwise_rel_p99"] < 1e-3
    # the float64 instantiation of the reference's templates (oracle/ref_harness.cpp *_f64) agrees with the float32 loops at
    # this size, and with the float64 autograd oracle much more closely
    got64, grads64, _sec, ctx64 = ref_net.forward_backward(net, coords, feats, threads=2, masks=masks, dtype=np.float64)
    assert got64.dtype == np.float64 and all(g.dtype == np.float64 for g in grads64.values())
    res64 = bench.compare_passes(dev, got64, [("%s.%s" % (type(mod).__name__, name), grads64[(id(mod), name)])
                                              for mod in net.modules() for name, p in mod._parameters.items() if p is not None],
                                 ctx64.mask_flips, ctx64.mask_total)
    assert res64["ok"], res64
    out64, stats64 = torch_ref.ref_forward(net, coords, feats, torch.float64, masks={id(mod): torch.from_numpy(mk) for mod, mk in zip(bns, dev["masks"])})
    # (not 1e-12: the reference computes invstd with powf in EVERY instantiation, sparseconvnet/SCN/CPU/BatchNormalization.cpp:40,
    #  so a float rounding per batch norm survives the double instantiation)
    assert np.abs(out64.detach().numpy() - got64).max() <= 2e-6 * np.abs(got64).max()
    # a wrong gradient is found and named
    name, shape, g = dev["grads"][7]
    bad = g.copy()
    bad.flat[0] += 10 * np.abs(g).max()
    dev["grads"][7] = (name, shape, bad)
    res = bench.compare_passes(dev, got, order, 0, 1)
    assert not res["ok"] and res["worst_tensor"].startswith(name) and "tensor 8 of" in res["worst_tensor"]


@pytest.mark.parametrize("name", ["dense", "sparse", "bernoulli", "word-edges"])
def test_occupancy_bitmap_algorithm_reproduces_the_reference_ranks(name):
    """WHY scn_batch_build_occ (round 6) agrees with the reference's sort + hash, on the CPU: oracle/occupancy.py restates the
    bitmap algorithm of csrc/scn_rules.hip in numpy (same box as scn_occ_plan -- compared field by field with the C function --,
    32 bits + a running count per word, the same index arithmetic) and reproduces what oracle/rules.py restates from the
    reference: site order, the site of every input row, the 27-neighbour table, the coarse grids and the fine -> coarse map of
    two coarsenings"""
    import ctypes
    from occuseg_amd import _lib, scenes
    from oracle import occupancy as O, rules as R
    coords = {"dense": lambda: random_points(0, 300, 2, 7),
              "sparse": lambda: random_points(1, 200, 3, 30, origin=10),
              "bernoulli": lambda: scenes.bernoulli_grid(12, 0.3, 3) + np.array([5, 1001, 37]),
              "word-edges": lambda: np.array([[x, y, 3, b] for b in (0, 1) for x in (30, 31, 32, 33, 62, 63, 64, 65, 96) for y in (7, 8)]
                                             + [[31, 8, 3, 1]], np.int64)}[name]()
    levels = 3
    occ = O.Occupancy(coords, levels)
    # the box: the same numbers as the C function
    c4 = coords if coords.shape[1] == 4 else np.concatenate([coords, np.zeros((len(coords), 1), np.int64)], 1)
    mm = (ctypes.c_int32 * 8)(*[int(v) for v in list(c4.min(0)) + list(c4.max(0))])
    box = _lib.OccBox()
    assert _lib.load().scn_occ_plan(mm, levels, ctypes.byref(box)) == 0
    assert list(box.origin) == [int(v) for v in occ.box.origin] and box.nb == occ.box.nb
    assert [list(box.n[l]) for l in range(levels)] == occ.box.n and list(box.wx)[:levels] == occ.box.wx
    assert list(box.woff)[:levels + 1] == occ.box.woff
    # scale 0: sites, rows, neighbours
    grid, header, table, site_of_point = R.input_layer_rules(coords)
    assert np.array_equal(occ.sites(0), grid.coords) and occ.n_sites[0] == grid.n
    assert np.array_equal(occ.site_of_row(), site_of_point)
    for l in range(levels):
        _, nbr = R.submanifold_rules(grid)
        assert np.array_equal(occ.neighbours(l), nbr), (name, l)
        if l + 1 < levels:
            coarse, rules, child = R.strided_rules(grid)
            assert np.array_equal(occ.sites(l + 1), coarse.coords) and occ.n_sites[l + 1] == coarse.n
            want = np.full(grid.n, -1, np.int64)
            for k in range(8):
                want[rules[k][:, 0]] = rules[k][:, 1]
            assert np.array_equal(occ.coarse_of_fine(l), want)
            grid = coarse
