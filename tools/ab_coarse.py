#!/usr/bin/env python3
"""Coarse UNet levels (launches that split their offsets): k_conv_tile2 against k_conv_tile4, both adding their partial tiles
themselves, shape by shape on the bench scenes -- forward and dInput of the submanifold and strided convolutions of levels 2..5,
alone on the chip, outputs compared bitwise.
usage: python tools/ab_coarse.py [--levels 3,4,5] [--m 64] [--policy 1500,3000]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes


def timeit(fn, iters=20, warm=3):
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
    ap.add_argument("--levels", default="3,4,5")
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--policy", default="1500", help="comma list: workgroups a split launch aims for (tuning key 4)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    coords, _ = scenes.batch([1000 + i for i in range(args.scenes)], 150000)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    levels = [int(x) for x in args.levels.split(",")]
    tot = {}
    for lv in range(max(levels) + 1):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        out_size = size // 2
        rbs = md.getRuleBook(size, out_size, f2, f2)
        n_c = md.getNActive(out_size)
        if lv in levels:
            p = args.m * (lv + 1)
            cases = [("sub %d->%d" % (p, p), rb, 27, p, p, n, n, 0, 1, False),
                     ("sub %d->%d" % (2 * p, p), rb, 27, 2 * p, p, n, n, 0, 1, False),
                     ("sub dI %d->%d" % (p, 2 * p), rb, 27, p, 2 * p, n, n, 0, 1, True)]
            if lv > 0:
                pass
            cases.append(("strided %d->%d" % (p, p + args.m), rbs, 8, p, p + args.m, n, n_c, 0, 1, False))
            for name, book, K, ci, co, n_in, n_out, ic, oc, tr in cases:
                x = torch.randn(n_in, ci, device=dev)
                w = torch.randn(K, co, ci, device=dev) * 0.05 if tr else torch.randn(K, ci, co, device=dev) * 0.05
                line = "L%d N=%d %-18s" % (lv, n_out, name)
                ref = None
                for pol in args.policy.split(","):
                    for t7 in (0, 1):
                        lib.scn_set_tuning(4, int(pol))
                        lib.scn_set_tuning(7, t7)
                        out = torch.empty(0, device=dev)
                        stats = []
                        fn = lambda: SCN._conv(x, out, w, None, book, ic, oc, n_out, tr, tr and K == 27, stats_out=stats)
                        fn()
                        code = lib.scn_conv_plan(K, book.stride, n_in, n_out, ci, co, 0)
                        if ref is None:
                            ref = out.clone()
                        same = torch.equal(ref, out)
                        ms = timeit(fn)
                        flops = 2.0 * book.total_pairs * ci * co
                        line += "  | pol %s plan %d: %.1f us %.1f TF%s" % (pol, code, ms * 1e3, flops / ms / 1e9, "" if same else " DIFFERS")
                        tot[(pol, t7)] = tot.get((pol, t7), 0.0) + ms
                print(line, flush=True)
        size = out_size
    lib.scn_set_tuning(7, 0)
    lib.scn_set_tuning(4, 1500)
    for k, v in sorted(tot.items()):
        print("policy %s tune7=%d: sum %.3f ms" % (k[0], k[1], v))
    print("stream-K / split wait gave up:", SCN.stream_k_gave_up())


if __name__ == "__main__":
    main()
