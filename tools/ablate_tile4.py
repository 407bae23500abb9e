#!/usr/bin/env python3
"""Timing-only ablation builds of k_conv_tile4 at 2 and at 1 wave per SIMD (tuning tool; results of ablated builds are wrong).
usage: python tools/ablate_tile4.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes
from tools.ab_tile6 import timeit

NAMES = {0: "full", 1: "no tile update", 2: "+ no row gathers", 8: "+ no weight loads", 4096: "no gathers/update/index loads",
         4097: "+ no weight loads (skeleton + MFMAs)", 2048: "wave 1 gathers nothing"}


def main():
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    coords, _ = scenes.batch([1000 + i for i in range(4)], 150000)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    rb = md.getSubmanifoldRuleBook(size, torch.tensor([3, 3, 3]))
    n = md.getNActive(size)
    for (ci, co) in ((64, 64), (128, 64)):
        x = torch.randn(n, ci, device=dev)
        w = torch.randn(27, ci, co, device=dev) * 0.05
        out = torch.empty(0, device=dev)
        flops = 2.0 * rb.total_pairs * ci * co
        for xl in (0, 15200):
            lib.scn_set_tuning(10, xl)
            for dbg in (0, 1, 2, 8, 4096, 4097, 2048, 0):
                lib.scn_set_tuning(1, dbg)
                ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                print("L0 %d->%d  %s waves/SIMD  %-40s %.3f ms %6.1f TF" % (ci, co, "1" if xl else "2", NAMES[dbg], ms, flops / ms / 1e9), flush=True)
    lib.scn_set_tuning(1, 0)
    lib.scn_set_tuning(10, 0)


if __name__ == "__main__":
    main()
