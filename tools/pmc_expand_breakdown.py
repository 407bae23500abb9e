#!/usr/bin/env python3
"""HBM (fabric) bytes of the k_conv_expand launches of a train step BY KIND -- deconvolution forward, strided-convolution dInput,
NetworkInNetwork forward / dInput -- from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py and the launch list
bench.py --dump-launches wrote in the same run (round-3 verdict, item 5: "find the 1.76 x").
usage: pmc_expand_breakdown.py <fetch_dir> <write_dir> <launches.json>"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "k_conv_expand" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    dump = json.load(open(sys.argv[3]))["in_order"]
    shapes = [tuple(l["shape"]) for l in dump if l["kernel"] == "conv_expand"]
    per_step = 20
    assert len(shapes) % per_step == 0 and len(fetch) % per_step == 0 and len(fetch) == len(write), (len(shapes), len(fetch), len(write))
    # position in the step -> (K, Ci, Co, transposed) (the same in every step; the row count alternates with the batch)
    kinds = [s for s in shapes[:per_step]]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
    for i, (f, w) in enumerate(zip(fetch, write)):
        K, n, ci, co, tr, incol = kinds[i % per_step]
        name = ("NiN" if K == 1 else "deconvolution" if not tr else "strided-conv dInput") + (" dInput" if (K == 1 and tr) else " forward" if K == 1 else "")
        key = (name, ci, co)
        a = agg[key]
        a[0] += 1
        a[1] += 2.0 * 1024.0 * f  # gfx950 correction x 2, KiB -> bytes (MI355X_MICROARCH.md, HBM section)
        a[2] += 1024.0 * w
        # algorithmic bytes of the launch: rows gathered + rows written + rules + weights; + what is fused into it:
        # a dInput launch also reads the batch norm's x and the other gradient of its output (n x Co each)
        a[3] += 4.0 * n * (ci + co) + 8.0 * n + 4.0 * K * ci * co
        a[4] += 4.0 * n * co * (2 if tr else 0)
    print("%-28s %9s %4s  %10s %10s %12s %12s  %s" % ("kind", "planes", "n", "fetch MB", "write MB", "algorithmic", "+ fused", "counted / (algorithmic + fused)"))
    for (name, ci, co), (c, f, w, al, fu) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %4d->%-4d %4d  %10.1f %10.1f %12.1f %12.1f  %.2f" % (name, ci, co, c, f / c / 1e6, w / c / 1e6, al / c / 1e6, (al + fu) / c / 1e6,
                                                                       (f + w) / (al + fu)))


if __name__ == "__main__":
    main()
