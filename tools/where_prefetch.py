#!/usr/bin/env python3
"""Where in the train step the next batch's rule build runs: for every rule-build kernel of a rocprofv3 --kernel-trace, the
index (in issue order, within its step) of the gather-mm / dWeight launch it overlaps.  usage: where_prefetch.py <dir>"""
import collections
import csv
import glob
import sys

iv = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
mm = [(s, e, n) for s, e, n in iv if "k_conv_tile4" in n or "k_dw_tile" in n or "k_conv_expand" in n or "k_conv_tile2" in n]
# steps: split the matrix launches at the k_conv_generic (stem forward) launches
stems = [s for s, e, n in iv if "k_conv_generic" in n or "k_conv_stem" in n]
hist = collections.Counter()
for s, e, n in iv:
    if "k_sub_probe" in n or "k_rules_fill" in n or "k_str_fill" in n or "k_hash_insert" in n:
        st = [t for t in stems if t <= s]
        if not st:
            continue
        t0 = st[-1]
        idx = sum(1 for a, b, _ in mm if t0 <= a <= s)
        tot = sum(1 for a, b, _ in mm if t0 <= a < (min([t for t in stems if t > t0] or [10**30])))
        hist[(idx * 10) // max(tot, 1)] += 1
print("rule-build kernels by decile of the step's matrix launches they run beside (0 = start of forward, 9 = end of backward):")
print(sorted(hist.items()))
