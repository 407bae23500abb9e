#!/usr/bin/env python3
"""k_conv_tile4 / k_dw_tile on warm and on cold caches: the same launch repeated back to back, and with 1 GiB written
between two launches (evicts L2 and the 256 MB Infinity Cache).  Inside a train step a launch finds neither its rulebook nor
its input rows in a cache: bench.py's `layers` show 88 TFLOP/s where tools/bench_conv.py shows 107.
usage: python tools/cold_cache.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes


def timed(fn, flush, iters=8):
    junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if flush else None
    ms = []
    for _ in range(iters + 2):
        if flush:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms = sorted(ms[2:])
    return ms[len(ms) // 2]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    for lv in range(3):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        p = 64 * (lv + 1)
        for ci, co in ((p, p), (2 * p, p)):
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            dy = torch.randn(n, co, device=dev)
            dw = torch.zeros_like(w)
            out = torch.empty(0, device=dev)
            flops = 2.0 * rb.total_pairs * ci * co

            def conv():
                SCN._conv(x, out, w, None, rb, 0, 1, n)

            def dweight():
                _lib.check(lib.scn_conv_dweight(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), rb.rules.data_ptr(),
                                                rb.counts.data_ptr(), None, 27, rb.stride, 0, 1, n, n, ci, co, 0, 0, None, 0, None, _lib.stream()))
            conv()
            line = "L%d %d->%d:" % (lv, ci, co)
            for name, fn in (("conv", conv), ("dW", dweight)):
                warm, cold = timed(fn, False), timed(fn, True)
                line += "  %s warm %.3f ms %.1f TF | cold %.3f ms %.1f TF (%+.0f %%)" % (
                    name, warm, flops / warm / 1e9, cold, flops / cold / 1e9, 100 * (cold / warm - 1))
            print(line, flush=True)
        md.getRuleBook(size, size // 2, f2, f2)
        size = size // 2


if __name__ == "__main__":
    main()
