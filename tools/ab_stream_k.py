#!/usr/bin/env python3
"""Stream-K against one workgroup per tile for k_conv_tile4 (scn_conv_forward's tile_passes / sk_state), per layer shape of the
bench scenes: kernel time, and every output compared bitwise with the one-workgroup-per-tile launch.
usage: python tools/ab_stream_k.py [--levels 0,1,2] [--sk 0,1,2,4] [--scenes 4] [--residual]"""
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
    ap.add_argument("--levels", default="0,1,2")
    ap.add_argument("--sk", default="0,1,2,4")
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--first-scene", type=int, default=1000)
    ap.add_argument("--sites", type=int, default=150000)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--min-items", type=int, default=512)
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--prio", type=int, default=1, help="0: stream-K launches without the alternating wave priority")
    ap.add_argument("--trunc", type=int, default=0, help="compute only the first TRUNC (tile, column block) items' worth of output rows")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    lib.scn_set_tuning(17, args.min_items)
    lib.scn_set_tuning(18, 0)  # (every eligible launch, whatever its round count)
    lib.scn_set_tuning(20, args.prio)
    SCN.STREAM_K = True  # (hand tile_passes and sk_state to every launch: tuning key 16 then decides)
    coords, _ = scenes.batch([args.first_scene + i for i in range(args.scenes)], args.sites)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    levels = [int(x) for x in args.levels.split(",")]
    sks = [int(x) for x in args.sk.split(",")]
    for lv in range(max(levels) + 1):
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        if lv in levels:
            p = args.m * (lv + 1)
            if args.trunc:  # (whole rounds of workgroups: no tail to balance, what is left is the cost of the persistent loop)
                n = min(n, args.trunc // (p // 64) * 128)
                md._tile_bases(rb, 1, n)
            tiles = (n + 127) // 128
            for (ci, co) in ((p, p), (2 * p, p)):
                torch.manual_seed(lv)
                x = torch.randn(n, ci, device=dev)
                w = torch.randn(27, ci, co, device=dev) * 0.05
                res = torch.randn(n, co, device=dev) if args.residual else None
                flops = 2.0 * rb.total_pairs * ci * co
                line = "L%d N=%d tiles=%d x%d nu=%.1f SC %d->%d GF=%.1f:" % (lv, n, tiles, co // 64, rb.total_pairs / n, ci, co, flops / 1e9)
                ref = None
                for sk in sks:
                    lib.scn_set_tuning(16, sk)
                    out = torch.empty(0, device=dev)
                    stats = []
                    fn = lambda: SCN._conv(x, out, w, None, rb, 0, 1, n, residual=res, stats_out=stats)
                    ms = min(timeit(fn) for _ in range(3))
                    same = ""
                    if ref is None:
                        ref = (out.clone(), stats[-1].clone())
                    else:
                        same = " bitwise=%s/%s" % (torch.equal(out, ref[0]), torch.equal(stats[-1], ref[1]))
                    if sk:
                        import ctypes
                        buf = (ctypes.c_ulonglong * 8)()
                        lib.scn_debug_stamps(buf)
                        fn()
                        torch.cuda.synchronize()
                        lib.scn_debug_stamps(buf)
                        same += " [seeded pieces %d, polls %d]" % (buf[7], buf[6])
                    st = SCN._sk_state(dev)
                    zero = int(st.view(torch.int32).abs().sum().item()) == 0
                    line += "  sk%d %.3f ms %.1f TF%s%s" % (sk, ms, flops / ms / 1e9, same, "" if zero else " STATE-NOT-ZERO")
                print(line, flush=True)
        md.getRuleBook(size, size // 2, f2, f2)
        size = size // 2


if __name__ == "__main__":
    main()
