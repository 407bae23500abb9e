#!/usr/bin/env python3
"""What the fused prologues / epilogues add to a k_conv_tile4 launch (inside the train step the forward launches of the
128->128 layers run at 86 TFLOP/s, the same launch alone at 106): the same gather-mm launch plain, with the column
statistics epilogue, with the residual prologue, with both, and as a dInput launch with the batch-norm backward epilogue.
usage: python tools/epilogue_cost.py [first scene seed, default 1100]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, scenes


def timed(fn, iters=12):
    for _ in range(3):
        fn()
    ms = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
    dev = torch.device("cuda:0")
    coords, _ = scenes.batch([seed + i for i in range(4)], 150000)
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
            wt = torch.randn(27, co, ci, device=dev) * 0.05  # dInput: a [K, co, ci] weight used transposed -> ci x co launch
            res = torch.randn(n, co, device=dev)
            out = torch.empty(0, device=dev)
            bx = torch.randn(n, co, device=dev)
            mean, invstd = bx.mean(0), 1.0 / bx.std(0)
            gamma, beta = torch.ones(co, device=dev), torch.zeros(co, device=dev)
            flops = 2.0 * rb.total_pairs * ci * co
            cases = (
                ("plain", lambda: SCN._conv(x, out, w, None, rb, 0, 1, n)),
                ("+stats", lambda: SCN._conv(x, out, w, None, rb, 0, 1, n, stats_out=[])),
                ("+residual", lambda: SCN._conv(x, out, w, None, rb, 0, 1, n, residual=res)),
                ("+residual+stats", lambda: SCN._conv(x, out, w, None, rb, 0, 1, n, residual=res, stats_out=[])),
                ("dInput", lambda: SCN._conv(x, out, wt, None, rb, 0, 1, n, w_transposed=True, w_reverse=True)),
                ("dInput+bn_bwd", lambda: SCN._conv(x, out, wt, None, rb, 0, 1, n, w_transposed=True, w_reverse=True,
                                                    bn_link=SCN.BnLink(bx, mean, invstd, gamma, beta, 0.0))),
            )
            line = "L%d N=%d %d->%d:" % (lv, n, ci, co)
            base = None
            for name, fn in cases:
                t = timed(fn)
                base = base or t
                line += "  %s %.3f ms %.1f TF (%+.0f %%)" % (name, t, flops / t / 1e9, 100 * (t / base - 1))
            print(line, flush=True)
        md.getRuleBook(size, size // 2, f2, f2)
        size = size // 2


if __name__ == "__main__":
    main()
