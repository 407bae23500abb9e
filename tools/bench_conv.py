#!/usr/bin/env python3
"""Microbenchmark of the gather-mm-scatter kernels on the bench scenes (tuning tool).
usage: python tools/bench_conv.py [--level L] [--variants 1,2] [--planes 64,128] [--scenes 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from occuseg_amd import SCN, _lib, scenes


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="0,1,2")
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--sites", type=int, default=150000)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--dw", action="store_true")
    ap.add_argument("--extra-lds", default="0", help="comma list: unused dynamic LDS bytes per k_conv_tile4 workgroup (occupancy probe)")
    ap.add_argument("--dw-variants", default="s0,0,s0,0")
    ap.add_argument("--asm", default="", help="comma list of tuning key 28 values: 0 = hipcc's pass loop, 1 = the assembly loop, 2.. = "
                                              "its timing-only ablations (tools/gen_tile4_loop.py ABLATIONS); stream-K off for these")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    coords, _ = scenes.batch([1000 + i for i in range(args.scenes)], args.sites)
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
            for (ci, co) in ((p, p), (2 * p, p)):
                x = torch.randn(n, ci, device=dev)
                w = torch.randn(27, ci, co, device=dev) * 0.05
                out = torch.empty(0, device=dev)
                flops = 2.0 * rb.total_pairs * ci * co
                line = "L%d N=%d nu=%.1f SC %d->%d GF=%.1f:" % (lv, n, rb.total_pairs / n, ci, co, flops / 1e9)
                for v, xl in [(v, xl) for v in args.variants.split(",") for xl in args.extra_lds.split(",")]:
                    lib.scn_set_tuning(10, int(xl))
                    lib.scn_set_tuning(0, int(v) % 10)
                    SCN.invalidate_weight_cache()  # variant 1 packs the weights in a different layout
                    lib.scn_set_tuning(1, (int(v) // 10) % 1000)
                    lib.scn_set_tuning(2, (int(v) // 10000) % 10)
                    lib.scn_set_tuning(4, int(v) // 100000)
                    ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                    line += "  v%s%s %.3f ms %.1f TF" % (v, "" if xl == "0" else "+%sB" % xl, ms, flops / ms / 1e9)
                for a in [int(t) for t in args.asm.split(",") if t]:
                    lib.scn_set_tuning(0, 4)
                    lib.scn_set_tuning(1, 0)
                    lib.scn_set_tuning(16, 0)
                    lib.scn_set_tuning(28, a)
                    ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                    line += "  asm%d %.3f ms %.1f TF" % (a, ms, flops / ms / 1e9)
                if args.asm:
                    lib.scn_set_tuning(28, 1)
                    lib.scn_set_tuning(16, 1)
                for dv in (args.dw_variants.split(',') if args.dw else []):
                    # "s<n>": the LDS-staged k_dw_tile (tuning key 12 = 0) with ablation n; plain numbers: default kernel choice
                    lib.scn_set_tuning(12, 0 if dv.startswith("s") else 1)
                    dv = int(dv.lstrip("s"))
                    lib.scn_set_tuning(6, dv % 1000)
                    lib.scn_set_tuning(5, dv // 1000 if dv >= 1000 else 256)
                    dy = torch.randn(n, co, device=dev)
                    dw = torch.zeros_like(w)
                    ms = timeit(lambda: _lib.check(lib.scn_conv_dweight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(),
                                rb.rules.data_ptr(), rb.counts.data_ptr(), None, 27, rb.stride, 0, 1, n, n, ci, co, 0, 0, None, 0, None, _lib.stream())))
                    line += "  dW[%d] %.3f ms %.1f TF" % (dv, ms, flops / ms / 1e9)
                lib.scn_set_tuning(6, 0)
                lib.scn_set_tuning(12, 1)
                print(line, flush=True)
        out_size = size // 2
        md.getRuleBook(size, out_size, f2, f2)
        size = out_size


if __name__ == "__main__":
    main()
