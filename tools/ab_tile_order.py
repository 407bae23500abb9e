#!/usr/bin/env python3
"""Dispatch order of the 128-row tiles of k_conv_tile4 (scn_conv_forward's tile_order): index order against "lightest
fraction last" and "longest first", per layer shape of the bench scenes, kernel time only.
usage: python tools/ab_tile_order.py [--levels 0,1,2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, scenes


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def orders(rb, n_out):
    bases = rb.bases[1]
    n_gran = bases.size(1) - 1
    n_tiles = (n_out + 127) // 128
    lo = torch.arange(n_tiles, device=bases.device) * 4
    hi = torch.clamp(lo + 4, max=n_gran)
    work = (bases[:, hi] - bases[:, lo]).sum(0)
    by_work = torch.argsort(work, descending=True, stable=True)
    res = {"index": None, "longest-first": by_work.int().contiguous()}
    for frac in (0.1, 0.25, 0.5):
        n_tail = int(n_tiles * frac)
        tail = by_work[n_tiles - n_tail:]
        keep = torch.ones(n_tiles, dtype=torch.bool, device=bases.device)
        keep[tail] = False
        res["light %.2f last" % frac] = torch.cat([torch.nonzero(keep).view(-1), tail]).int().contiguous()
        # light tiles last but in INDEX order among themselves (locality kept inside the tail too)
        res["light %.2f last, index order" % frac] = torch.cat([torch.nonzero(keep).view(-1), torch.sort(tail).values]).int().contiguous()
    w = work.float()
    return res, (float(w.mean()), float(w.min()), float(w.max()), float(w.std()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="0,1,2")
    ap.add_argument("--m", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    SCN.TILE_ORDER = 0
    coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    levels = [int(x) for x in args.levels.split(",")]
    for lv in range(max(levels) + 1):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        if lv in levels:
            od, st = orders(rb, n)
            print("level %d: %d sites, %d tiles, pairs per tile mean %.0f min %.0f max %.0f std %.0f" % ((lv, n, (n + 127) // 128) + st))
            p = args.m * (lv + 1)
            for (ci, co) in ((p, p), (2 * p, p)):
                x = torch.randn(n, ci, device=dev)
                w = torch.randn(27, ci, co, device=dev) * 0.05
                out = torch.empty(0, device=dev)
                flops = 2.0 * rb.total_pairs * ci * co
                line = []
                for name, o in od.items():
                    if o is None:
                        rb.order.pop(1, None)
                    else:
                        rb.order[1] = o
                    ms = min(timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n)) for _ in range(3))
                    line.append("%s %.3f ms %.1f TF" % (name, ms, flops / ms / 1e9))
                rb.order.pop(1, None)
                print("  %d->%d: " % (ci, co) + " | ".join(line), flush=True)
        md.getRuleBook(size, size // 2, f2, f2)
        size = size // 2


if __name__ == "__main__":
    main()
