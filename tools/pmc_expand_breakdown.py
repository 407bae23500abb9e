#!/usr/bin/env python3
"""HBM (fabric) bytes of the k_conv_expand launches of a train step BY KIND -- deconvolution forward, strided-convolution dInput,
NetworkInNetwork forward / dInput -- from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py and the launch list
bench.py --dump-launches wrote in the same run (round-3 verdict, item 5: "find the 1.76 x").
usage: pmc_expand_breakdown.py <fetch_dir> <write_dir> <launches.json> [conv_expand | conv_tile4 | conv_tile4_split | dw_tile]"""
import collections
import csv
import glob
import json
import sys


def is_kernel(name, which):
    split = any(t in name for t in ("k_conv_tile4<0, 0, 1, 0, 1>", "k_conv_tile4<0, 1, 1, 0, 1>", "k_conv_tile4<0, 0, 1, 0, 1,", "k_conv_tile4<0, 1, 1, 0, 1,"))
    if which == "conv_tile4":
        return "k_conv_tile4" in name and not split
    if which == "conv_tile4_split":
        return split
    return {"conv_expand": "k_conv_expand", "dw_tile": "k_dw_tile"}[which] in name


def load(d, counter, which):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and is_kernel(r["Kernel_Name"], which)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    which = sys.argv[4] if len(sys.argv) > 4 else "conv_expand"
    fetch, write = load(sys.argv[1], "FETCH_SIZE", which), load(sys.argv[2], "WRITE_SIZE", which)
    dump = json.load(open(sys.argv[3]))["in_order"]
    entries = [l for l in dump if l["kernel"] == which]
    steps = 2  # (bench.py --steps 2 --event-every 1: the dump holds two steps, the trace those, the warm-up step and the extra one)
    per_step = len(entries) // steps
    assert per_step * steps == len(entries) and len(fetch) % per_step == 0 and len(fetch) == len(write), (len(entries), len(fetch), len(write))
    # position in the step -> shape, algorithmic bytes (the same kind in every step; the row count alternates with the batch)
    kinds = entries[:per_step]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
    for i, (f, w) in enumerate(zip(fetch, write)):
        e = kinds[i % per_step]
        K, n, ci, co, tr, incol = e["shape"]
        if which == "conv_expand":
            name = ("NiN" if K == 1 else "deconvolution" if not tr else "strided-conv dInput") + (" dInput" if (K == 1 and tr) else " forward" if K == 1 else "")
            algo = 4.0 * n * (ci + co) + 8.0 * n + 4.0 * K * ci * co   # every output row is one pair
            fused = 4.0 * n * co * (2 if tr else 0)  # a dInput launch also reads the batch norm's x and the other gradient of its output
        else:
            name = "%s K=%d%s" % (which, K, " dInput" if tr else "")
            algo = e.get("bytes", 0.0)   # (bench.py's own figure for the launch: SURVEY 8d)
            fused = 0.0
        a = agg[(name, ci, co)]
        a[0] += 1
        a[1] += 2.0 * 1024.0 * f  # gfx950 correction x 2, KiB -> bytes (MI355X_MICROARCH.md, HBM section)
        a[2] += 1024.0 * w
        a[3] += algo
        a[4] += fused
    print("%-28s %9s %4s  %10s %10s %12s %12s  %s" % ("kind", "planes", "n", "fetch MB", "write MB", "algorithmic", "+ fused", "counted / (algorithmic + fused)"))
    for (name, ci, co), (c, f, w, al, fu) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %4d->%-4d %4d  %10.1f %10.1f %12.1f %12.1f  %.2f" % (name, ci, co, c, f / c / 1e6, w / c / 1e6, al / c / 1e6, (al + fu) / c / 1e6,
                                                                       (f + w) / (al + fu)))


if __name__ == "__main__":
    main()
