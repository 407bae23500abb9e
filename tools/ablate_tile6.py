#!/usr/bin/env python3
"""Timing-only builds of k_conv_tile6<128> (and k_conv_tile4 beside it) at 2 and at 1 wave per SIMD (tuning tool).
usage: python tools/ablate_tile6.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes
from tools.ab_tile6 import timeit

NAMES = {0: "full", 1: "no tile update", 3: "no tile update, no row loads, no waits", 4: "no weight loads, no waits",
         14: "no loads, no waits", 15: "MFMAs + skeleton", 16: "all loads, no waits", 20: "no weight loads, no waits",
         32: "weight loads never masked", 64: "tile writes never masked", 96: "neither masked (EXEC never written)",
         512: "accumulators in AGPRs", 528: "accumulators in AGPRs, no waits",
         1040: "no waits, every reload one run later", 2064: "no waits, reloads in pairs that share cache lines",
         144: "no waits, half of the weight loads", 272: "no waits, weight loads of one dword per lane"}


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
        for xl in (0,):
            lib.scn_set_tuning(10, xl)
            lib.scn_set_tuning(11, 0)
            lib.scn_set_tuning(1, 0)
            ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
            print("L0 %d->%d  %s waves/SIMD  %-45s %.3f ms %6.1f TF" % (ci, co, "1" if xl else "2", "k_conv_tile4", ms, flops / ms / 1e9), flush=True)
            lib.scn_set_tuning(11, 128)
            for dbg in (0, 16, 2064, 16, 2064, 0):
                lib.scn_set_tuning(1, dbg)
                ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                print("L0 %d->%d  %s waves/SIMD  %-45s %.3f ms %6.1f TF" % (ci, co, "1" if xl else "2", "tile6: " + NAMES[dbg], ms, flops / ms / 1e9), flush=True)
    lib.scn_set_tuning(1, 0)
    lib.scn_set_tuning(10, 0)
    lib.scn_set_tuning(11, 0)


if __name__ == "__main__":
    main()
