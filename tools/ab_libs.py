#!/usr/bin/env python3
"""In-process A/B of TWO builds of the library (same ABI) on the bench scenes' gather-mm layers: TFLOP/s and
bit-equality of the outputs.  usage: python tools/ab_libs.py <libA.so> [<libB.so> = occuseg_amd/libscn_hip.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from occuseg_amd import SCN, _lib, scenes
from bench_conv import timeit


def main():
    path_a = os.path.abspath(sys.argv[1])
    path_b = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else _lib.LIB_PATH
    libs = {"A": _lib._open(path_a, _lib.SIGNATURES), "B": _lib._open(path_b, _lib.SIGNATURES)}
    print("A =", path_a, "\nB =", path_b, flush=True)
    dev = torch.device("cuda:0")
    _lib._lib = libs["B"]
    coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    m = 64
    for lv in range(3):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        p = m * (lv + 1)
        for (ci, co) in ((p, p), (2 * p, p), (p, 2 * p)):
            if ci > 320:
                continue
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            flops = 2.0 * rb.total_pairs * ci * co
            outs, line = {}, "L%d N=%d SC %d->%d:" % (lv, n, ci, co)
            for name in ("A", "B", "A", "B"):
                _lib._lib = libs[name]
                SCN.invalidate_weight_cache()
                out = torch.empty(0, device=dev)
                ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                outs[name] = out.clone()
                line += "  %s %.3f ms %.1f TF" % (name, ms, flops / ms / 1e9)
            print(line + "  equal=%s" % torch.equal(outs["A"], outs["B"]), flush=True)
        out_size = size // 2
        _lib._lib = libs["B"]
        md.getRuleBook(size, out_size, f2, f2)
        size = out_size


if __name__ == "__main__":
    main()
