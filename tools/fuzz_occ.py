#!/usr/bin/env python3
"""Randomised clouds through both batch builds -- sort + hash, one grid and rulebook at a time as the ops ask (what
tests/test_gpu_rules.py::test_rulebooks_bit_exact holds against the oracle) and occupancy bitmaps (scn_batch_build_occ) -- comparing
every grid, rulebook, tile-base table and the input layer's order / site_of_row / site_start bit for bit: random extents and
origins (word edges, unaligned minima), 1-6 scales, 1-5 samples whose indices need not start at 0, duplicates (a few voxels with
many rows), int32 / int64 / float32 coordinates with three or four columns.
usage: python tools/fuzz_occ.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from occuseg_amd import SCN


def walk(m, size, levels):
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    out = []
    for level in range(levels):
        rb = m.getSubmanifoldRuleBook(size, f3)
        g = m.grids[tuple(int(v) for v in size)]
        out += [g.coords, g.keys, rb.counts, rb.bases[1]] + [torch.from_numpy(a) for a in rb.host_lists()]
        if level + 1 < levels:
            rb2 = m.getRuleBook(size, size // 2, f2, f2)
            out += [rb2.counts, rb2.bases[0], rb2.bases[1]] + [torch.from_numpy(a) for a in rb2.host_lists()]
            size = size // 2
    return out + [m.input_order, m.input_site_of_row, m.input_site_start]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    SCN.GRID_CHAIN = False
    bad = 0
    for case in range(cases):
        levels = int(rng.integers(1, 7))
        nb = int(rng.integers(1, 6))
        ext = rng.integers(1, 140, size=3)
        origin = np.array([rng.integers(0, 1024 - ext[0]), rng.integers(0, 2048 - ext[1]), rng.integers(0, 1024 - ext[2])])
        P = int(rng.integers(1, 60000))
        c = rng.integers(0, ext, size=(P, 3)) + origin
        ncols = 4 if nb > 1 or rng.random() < 0.7 else 3
        if ncols == 4:
            b = np.sort(rng.integers(0, nb, size=P)) + int(rng.integers(0, 3))
            c = np.concatenate([c, b[:, None]], 1)
        if rng.random() < 0.3:  # a few voxels with many rows (up to beyond the segment sort's 64)
            for _ in range(int(rng.integers(1, 4))):
                rows = rng.permutation(P)[:int(rng.integers(2, 150))]
                c[rows, :3] = c[rows[0], :3]
                if ncols == 4:
                    c[rows, 3] = c[rows[0], 3]
            if ncols == 4:
                c = c[np.argsort(c[:, 3], kind="stable")]
        dtype = (torch.int64, torch.int32, torch.float32)[int(rng.integers(0, 3))]
        coords = torch.from_numpy(np.ascontiguousarray(c)).to(dtype)
        if dtype == torch.float32:
            coords = coords + float(rng.random() * 0.9)
        size = torch.tensor([4096 * 2 ** int(rng.integers(0, 2))] * 3, dtype=torch.int64)
        res = []
        for occ in (False, True):
            SCN.OCC_BUILD = occ
            SCN._PLANS.clear()
            before = SCN.OCC_STATS["built"]
            m = SCN.Metadata_3()
            m.inputLayer(size, coords, 0, 4, dev)
            res.append(walk(m, size, levels))
            if occ:  # ... and once more with the plan the walk recorded (exactly the scales and rulebooks it asked for)
                assert SCN.OCC_STATS["built"] == before + 1, "the bitmap build did not run"
                m = SCN.Metadata_3()
                m.inputLayer(size, coords, 0, 4, dev)
                res.append(walk(m, size, levels))
                assert SCN.OCC_STATS["built"] == before + 2
        ok = all(len(res[0]) == len(r) and all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(res[0], r)) for r in res[1:])
        if not ok:
            bad += 1
            print("MISMATCH case %d: levels %d samples %d extent %s origin %s P %d ncols %d %s" % (case, levels, nb, ext, origin, P, ncols, dtype))
    print("fuzz_occ: %d cases, %d mismatches (bitmap build -- first batch and replayed plan -- against the sort + hash build); bitmap "
          "builds %d, host repairs of long segments %d" % (cases, bad, SCN.OCC_STATS["built"], SCN.OCC_STATS["order_repaired"]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
