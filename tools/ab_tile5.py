#!/usr/bin/env python3
"""A/B of k_conv_tile4 (two waves x 32 planes) against k_conv_tile5 (one wave x 64 planes) on the bench scenes:
bit-equality of the outputs and TFLOP/s per layer shape.  usage: python tools/ab_tile5.py [--levels 0,1,2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes
from bench_conv import timeit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="0,1,2")
    ap.add_argument("--m", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
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
            p = args.m * (lv + 1)
            for (ci, co) in ((p, p), (2 * p, p), (p, 2 * p)):
                if ci > 320:
                    continue
                x = torch.randn(n, ci, device=dev)
                w = torch.randn(27, ci, co, device=dev) * 0.05
                flops = 2.0 * rb.total_pairs * ci * co
                outs, line = {}, "L%d N=%d SC %d->%d:" % (lv, n, ci, co)
                for v in (0, 1, 0, 1):
                    lib.scn_set_tuning(9, v)
                    out = torch.empty(0, device=dev)
                    ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                    outs[v] = out.clone()
                    line += "  tile%d %.3f ms %.1f TF" % (5 if v else 4, ms, flops / ms / 1e9)
                lib.scn_set_tuning(9, 0)
                line += "  equal=%s finite=%s" % (torch.equal(outs[0], outs[1]), bool(torch.isfinite(outs[1]).all()))
                print(line, flush=True)
        out_size = size // 2
        md.getRuleBook(size, out_size, f2, f2)
        size = out_size


if __name__ == "__main__":
    main()
