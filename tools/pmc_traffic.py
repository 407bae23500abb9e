#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into per-launch HBM traffic of
the gather-mm kernels.  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -> doubled."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = "conv_tile" if "k_conv_tile" in name else "dw_tile" if "k_dw_tile" in name else None
        if key:
            agg[key].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    n = len(fetch[k])
    fb = 2.0 * 1024.0 * sum(fetch[k]) / n          # gfx950 correction x2, KiB -> bytes
    wb = 1024.0 * sum(write[k]) / max(len(write[k]), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
