#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into per-launch HBM traffic of
the matrix kernels.  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -> doubled.
The output is stamped with the hash of the kernel sources it was measured on (bench.source_hash):
bench.py refuses a traffic file that belongs to other sources."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (the offset-split instantiation of k_conv_tile4 -- template arguments <DBG, INCOL, GT, SK, SPLIT = 1>: the coarse levels -- is a
#  kernel of its own in bench.py's tables; since round 6 a sixth argument, ASM, follows)
KERNELS = (("k_conv_tile4<0, 0, 1, 0, 1,", "conv_tile4_split"), ("k_conv_tile4<0, 1, 1, 0, 1,", "conv_tile4_split"), ("k_conv_tile4<0, 0, 1, 0, 1>", "conv_tile4_split"), ("k_conv_tile4<0, 1, 1, 0, 1>", "conv_tile4_split"), ("k_conv_tile4", "conv_tile4"), ("k_conv_tile2", "conv_tile2"), ("k_conv_expand", "conv_expand"),
           ("k_conv_tile<", "conv_tile_fallback"), ("k_conv_generic", "conv_generic"), ("k_conv_stem", "conv_stem"), ("k_dw_stem", "dw_stem"), ("k_dw_tile", "dw_tile"),
           ("k_bn_", "bn"), ("k_dw_narrow", "dw_narrow"))


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = next((k for pat, k in KERNELS if pat in name), None)
        if key:
            agg[key].append(float(r["Counter_Value"]))
    return agg


def main():
    import bench
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"source_hash": bench.source_hash(),
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py "
                  "--no-cpu-baseline`; KiB -> bytes; FETCH_SIZE x 2 (gfx950 note of the microarch guide)"}
    for k in fetch:
        n = len(fetch[k])
        fb = 2.0 * 1024.0 * sum(fetch[k]) / n          # gfx950 correction x2, KiB -> bytes
        wb = 1024.0 * sum(write[k]) / max(len(write[k]), 1)
        out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
