#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run in the build container (needs /root/reference for oracle/_ref):

    make -C oracle && python tests/golden/make_golden.py

Fixtures are DATA: seeded inputs plus expected outputs.  Rulebooks / site order come from
oracle.rules (cross-checked against oracle.bruteforce before being written; the reference has no
runnable code or vectors for them: parity unpinned, see oracle/__init__.py); all floating-point
expectations come from the reference's own CPU loops compiled into oracle/_ref (oracle/ref.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from occuseg_amd import scenes  # noqa: E402
from oracle import bruteforce as BF, ref, rules as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rules_fixture(name, coords):
    grid, header, table, sop = R.input_layer_rules(coords)
    sc, ctr, rows, ma = BF.sites_and_input_rules([tuple(int(v) for v in (list(r) + [0])[:4]) for r in coords.tolist()])
    assert [tuple(r) for r in grid.coords.tolist()] == sc and header[1] == ma
    sub, _ = R.submanifold_rules(grid)
    bsub = BF.submanifold_rules(sc)
    for k in range(27):
        assert [tuple(x) for x in sub[k].tolist()] == bsub[k]
    coarse, strd, _ = R.strided_rules(grid)
    cc, cctr, bstr = BF.strided_rules(sc)
    assert [tuple(r) for r in coarse.coords.tolist()] == cc
    for k in range(8):
        assert [tuple(x) for x in strd[k].tolist()] == bstr[k]
    d = dict(coords=coords, site_coords=grid.coords.astype(np.int32), header=header, table=table,
             site_of_point=sop.astype(np.int32), coarse_coords=coarse.coords.astype(np.int32))
    for k in range(27):
        d["sub%02d" % k] = sub[k]
    for k in range(8):
        d["str%d" % k] = strd[k]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    return grid, sub, coarse, strd, table


def main():
    assert ref.available(), "build oracle/_ref first: make -C oracle"
    rng = np.random.default_rng(2024)
    # G1: 8^3 grid, 2 samples, hand-placed edge sites and duplicates
    edge = [[0, 0, 0, 0], [0, 0, 0, 0], [7, 7, 7, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [3, 3, 3, 0],
            [3, 3, 4, 0], [3, 4, 3, 0], [4, 3, 3, 0], [3, 3, 3, 0], [6, 6, 6, 0], [6, 6, 7, 0]]
    edge += [[int(v) for v in rng.integers(0, 8, 3)] + [0] for _ in range(14)]
    edge += [[0, 0, 0, 1], [7, 0, 7, 1], [2, 2, 2, 1], [2, 2, 2, 1], [2, 3, 2, 1]]
    edge += [[int(v) for v in rng.integers(0, 8, 3)] + [1] for _ in range(8)]
    rules_fixture("g1_rules_8cube", np.array(edge, np.int64))

    # G2: configs[0] site set -- 32^3 Bernoulli(0.25) seed 0: rulebooks only (integers compress well)
    rules_fixture("g2_rules_bernoulli32", scenes.bernoulli_grid(32, 0.25, 0))
    # G3: same recipe on a 14^3 grid (keeps the float fixture < 1 MB): SC 16->16 fwd/bwd, conv 16->32, deconv 32->16
    coords = scenes.bernoulli_grid(14, 0.25, 0, offset=5)
    grid, sub, coarse, strd, table = rules_fixture("g3_rules_bernoulli14", coords)
    n, nc = grid.n, coarse.n
    x = rng.normal(size=(n, 16)).astype(np.float32)
    w = (rng.normal(size=(27, 16, 16)) * (2.0 / 16 / 27) ** 0.5).astype(np.float32)
    dy = rng.normal(size=(n, 16)).astype(np.float32)
    y = ref.conv_forward(x, w, sub, n)
    dx, dw = ref.conv_backward(x, dy, w, sub)
    wc = (rng.normal(size=(8, 16, 32)) * (2.0 / 16 / 8) ** 0.5).astype(np.float32)
    dyc = rng.normal(size=(nc, 32)).astype(np.float32)
    yc = ref.conv_forward(x, wc, strd, nc)
    dxc, dwc = ref.conv_backward(x, dyc, wc, strd)
    xd = rng.normal(size=(nc, 32)).astype(np.float32)
    wd = (rng.normal(size=(8, 32, 16)) * (2.0 / 32 / 8) ** 0.5).astype(np.float32)
    dyd = rng.normal(size=(n, 16)).astype(np.float32)
    yd = ref.conv_forward(xd, wd, strd, n, in_col=1)
    dxd, dwd = ref.conv_backward(xd, dyd, wd, strd, in_col=1)
    np.savez_compressed(os.path.join(OUT, "g3_conv_bernoulli14.npz"), x=x, w=w, dy=dy, y=y, dx=dx, dw=dw, wc=wc,
                        dyc=dyc, yc=yc, dxc=dxc, dwc=dwc, xd=xd, wd=wd, dyd=dyd, yd=yd, dxd=dxd, dwd=dwd)

    # G4: batch norm, C=16, N=1024 and the N=2 edge (N-1 divisor), train + eval, three leakiness values
    d = {}
    for nn in (1024, 2):
        xb = (rng.normal(size=(nn, 16)) * rng.uniform(0.5, 2, 16) + rng.normal(size=16)).astype(np.float32)
        g = rng.uniform(0.5, 1.5, 16).astype(np.float32)
        b = rng.normal(size=16).astype(np.float32)
        rm0 = rng.normal(size=16).astype(np.float32)
        rv0 = rng.uniform(0.5, 2, 16).astype(np.float32)
        dyb = rng.normal(size=(nn, 16)).astype(np.float32)
        d.update({"x%d" % nn: xb, "g%d" % nn: g, "b%d" % nn: b, "rm%d" % nn: rm0, "rv%d" % nn: rv0, "dy%d" % nn: dyb})
        for li, leak in enumerate((0.0, 0.333, 1.0)):
            for train in (1, 0):
                yb, sm, si, rm, rv = ref.bn_forward(xb, g, b, rm0, rv0, 1e-4, 0.9, bool(train), leak)
                tag = "%d_l%d_t%d" % (nn, li, train)
                d.update({"y" + tag: yb, "sm" + tag: sm, "si" + tag: si, "rmo" + tag: rm, "rvo" + tag: rv})
                if train:
                    dxb, dmask, dwb, dbb = ref.bn_backward(xb, yb, dyb, sm, si, g, leak)
                    d.update({"dx" + tag: dxb, "dm" + tag: dmask, "dw" + tag: dwb, "db" + tag: dbb})
    np.savez_compressed(os.path.join(OUT, "g4_batchnorm.npz"), **d)

    # G5: input / output layer on the G1 sites (modes 3 and 4)
    coords = np.array(edge, np.int64)
    _, _, table, _ = R.input_layer_rules(coords)
    f = rng.normal(size=(len(coords), 3)).astype(np.float32)
    dd = dict(feats=f, table=table)
    for mode in (3, 4):
        o = ref.input_forward(f, table, mode)
        dd["out_m%d" % mode] = o
        dd["din_m%d" % mode] = ref.input_backward(o, table, len(coords), mode)
    np.savez_compressed(os.path.join(OUT, "g5_io_layers.npz"), **dd)
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
