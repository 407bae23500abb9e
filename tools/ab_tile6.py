#!/usr/bin/env python3
"""k_conv_tile6 (three waves per SIMD) against k_conv_tile4 on the bench scenes: results, then times (tuning tool).
usage: python tools/ab_tile6.py [--sites 150000] [--scenes 4] [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    ap.add_argument("--sites", type=int, default=150000)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--rows", default="0,128,96")
    ap.add_argument("--extra-lds", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    coords, _ = scenes.batch([1000 + i for i in range(args.scenes)], args.sites)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    variants = [int(v) for v in args.rows.split(",")]
    lib.scn_set_tuning(10, args.extra_lds)
    bad = 0
    for lv in range(2):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        p = 64 * (lv + 1)
        for (ci, co) in ((p, p), (2 * p, p)):
            x = torch.randn(n, ci, device=dev)
            w = torch.randn(27, ci, co, device=dev) * 0.05
            b = torch.randn(co, device=dev)
            res = torch.randn(n, co, device=dev)
            flops = 2.0 * rb.total_pairs * ci * co
            outs, stats = {}, {}
            for v in variants:
                lib.scn_set_tuning(11, v)
                for mode in ("fwd", "dinput", "residual+stats"):
                    out = torch.empty(0, device=dev)
                    if mode == "fwd":
                        SCN._conv(x, out, w, b, rb, 0, 1, n)
                    elif mode == "dinput":
                        SCN._conv(torch.randn(n, co, device=dev, generator=torch.Generator(dev).manual_seed(7)), out, w, None, rb, 0, 1, n, True, True)
                    else:
                        sink = []
                        SCN._conv(x, out, w, None, rb, 0, 1, n, residual=res, stats_out=sink)
                        # column sums / sums of squares over all tiles, whatever the tile height
                        stats[v] = sink[0].double().sum(0) if sink else None
                    outs[(v, mode)] = out.clone()
            for v in variants[1:]:
                for mode in ("fwd", "dinput", "residual+stats"):
                    ref, o = outs[(variants[0], mode)], outs[(v, mode)]
                    err = (o - ref).abs().max().item() / ref.abs().max().item()
                    ok = err < 1e-5
                    if mode == "residual+stats":
                        full = torch.stack([o.double().sum(0), (o.double() ** 2).sum(0)])
                        serr = ((stats[v] - full).abs().max() / full.abs().max()).item()
                        ok = ok and serr < 1e-5
                        mode += " (stats err %.2g)" % serr
                    bad += not ok
                    print("L%d %d->%d rows=%d vs %d %s: rel err %.3g %s" % (lv, ci, co, v, variants[0], mode, err, "OK" if ok else "MISMATCH"), flush=True)
            # run-to-run reproducibility of the new kernel
            lib.scn_set_tuning(11, variants[-1])
            o1, o2 = torch.empty(0, device=dev), torch.empty(0, device=dev)
            SCN._conv(x, o1, w, b, rb, 0, 1, n)
            SCN._conv(x, o2, w, b, rb, 0, 1, n)
            print("   repeat bitwise equal:", torch.equal(o1, o2), flush=True)
            bad += not torch.equal(o1, o2)
            line = "L%d N=%d nu=%.1f SC %d->%d:" % (lv, n, rb.total_pairs / n, ci, co)
            out = torch.empty(0, device=dev)
            for r in range(args.rounds):
                for v in variants:
                    lib.scn_set_tuning(11, v)
                    ms = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n))
                    line += "  [%d] %.3f ms %.1f TF" % (v, ms, flops / ms / 1e9)
            print(line, flush=True)
        out_size = size // 2
        md.getRuleBook(size, out_size, f2, f2)
        size = out_size
    lib.scn_set_tuning(11, 0)
    print("MISMATCHES:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
