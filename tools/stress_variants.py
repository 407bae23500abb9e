#!/usr/bin/env python3
"""Randomised cross-check of the two gather-mm kernels (k_conv_tile4 vs k_conv_tile2): random clouds of varying
density / extent / batch, random plane counts, submanifold and strided rulebooks in both directions, forward and
transposed weights.  Both kernels use the same summation order, so every output must be BIT-identical; any
scheduling hazard in the hand-written pipeline of k_conv_tile4 shows up as a difference.
usage: python tools/stress_variants.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from occuseg_amd import SCN, _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    ran4, bad = run(args.cases, args.seed)
    return 1 if bad else 0


def run(cases, seed, max_points=400000, verbose=True):
    """returns (cases that went through k_conv_tile4, mismatches); leaves the tuning build selected"""
    class args:  # noqa: N801
        pass
    args.cases, args.seed = cases, seed
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    lib.scn_set_tuning(3, 1)  # launches with few tiles: no offset split
    lib.scn_set_tuning(8, 0)  # deconvolution-direction launches through the tile kernels too (k_conv_expand keeps the whole
                              # C_i reduction in registers: another summation order, equal to rounding only)
    rng = np.random.default_rng(args.seed)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    size = torch.tensor([4096] * 3)
    bad = 0
    ran4 = 0
    for case in range(args.cases):
        batch = int(rng.integers(1, 4))
        extent = int(rng.integers(40, 120))
        fill = float(rng.uniform(0.08, 0.95))
        pts = []
        for b in range(batch):
            n = int(fill * extent ** 3 * rng.uniform(0.5, 1.0))
            n = max(min(n, max_points), 20000)
            c = rng.integers(0, extent, (n, 3)) + int(rng.integers(0, 50))
            pts.append(np.concatenate([c, np.full((n, 1), b)], 1))
        coords = torch.from_numpy(np.concatenate(pts).astype(np.int64))
        md = SCN.Metadata_3()
        md.inputLayer(size, coords, 0, 4, dev)
        n0 = md.getNActive(size)
        kind = int(rng.integers(0, 3))
        ci = int(rng.choice([64, 128, 192, 256, 320, 384]))
        co = int(rng.choice([64, 128, 192, 256]))
        if kind == 0:
            rb, K, in_col, out_col, n_in, n_out = md.getSubmanifoldRuleBook(size, f3), 27, 0, 1, n0, n0
        else:
            rb = md.getRuleBook(size, size // 2, f2, f2)
            n1 = md.getNActive(size // 2)
            K = 8
            in_col, out_col, n_in, n_out = (0, 1, n0, n1) if kind == 1 else (1, 0, n1, n0)
        wt, wr = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and K == 27
        x = torch.randn(n_in, ci, device=dev)
        w = torch.randn(K, co, ci, device=dev) * 0.05 if wt else torch.randn(K, ci, co, device=dev) * 0.05
        bias = torch.randn(co, device=dev) if rng.integers(0, 2) else None
        outs = []
        for v in (4, 2):
            lib.scn_set_tuning(0, v)
            SCN.invalidate_weight_cache()  # variant 1 packs the weights in a different layout
            out = torch.empty(0, device=dev)
            SCN._conv(x, out, w, bias, rb, in_col, out_col, n_out, w_transposed=wt, w_reverse=wr)
            outs.append(out.clone())
        lib.scn_set_tuning(0, 4)
        tiles = (n_out + 127) // 128 * (co // 64)
        ran4 += tiles >= 512
        same = torch.equal(outs[0], outs[1])
        finite = bool(torch.isfinite(outs[0]).all())
        if not same or not finite:
            bad += 1
            d = (outs[0] - outs[1]).abs()
            print("MISMATCH case %d: kind %d K %d n_in %d n_out %d %d->%d wt %d wr %d bias %d  max|d| %.3g rows %d"
                  % (case, kind, K, n_in, n_out, ci, co, wt, wr, bias is not None, float(d.max()),
                     int((d.max(1).values > 0).sum())), flush=True)
        if case % 20 == 0 and verbose:
            print("case %d: n_out %d tiles %d %d->%d K %d %s" % (case, n_out, tiles, ci, co, K, "ok" if same else "BAD"), flush=True)
    print("stress_variants: %d cases, %d through k_conv_tile4, %d mismatches" % (args.cases, ran4, bad))
    lib.scn_set_tuning(3, 0)
    lib.scn_set_tuning(8, 1)
    return ran4, bad


if __name__ == "__main__":
    sys.exit(main())
