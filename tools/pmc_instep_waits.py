#!/usr/bin/env python3
"""In-step wait counters per kernel (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace
of bench.py): the share of its wave-cycles a kernel's waves spend waiting on memory counters / on instruction issue.
usage: pmc_instep_waits.py <dir>"""
import collections
import csv
import glob
import sys

cnt = collections.defaultdict(lambda: collections.defaultdict(float))
key = lambda n: next((k for k in ("k_conv_tile4", "k_conv_tile2", "k_conv_expand", "k_dw_tile") if k in n), None)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(cnt.items()):
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        print("%-14s SQ_WAIT_ANY / SQ_WAVE_CYCLES %.3f   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES %.3f" % (k, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc))
