#!/usr/bin/env python3
"""Per-kernel clock and MFMA-pipe occupancy INSIDE the train step, from one rocprofv3 pass
(--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace) of bench.py.  usage: pmc_instep.py <dir>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
key = lambda n: next((k for k in ("k_conv_tile4", "k_conv_tile2", "k_conv_expand", "k_dw_tile", "k_bn_", "k_conv_generic", "k_conv_stem", "k_dw_stem", "k_dw_rows") if k in n), None)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
if not any(dur.values()):
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r["Kernel_Name"])
            if k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
print("# kernel: dispatches, mean us, clock GHz = GRBM_GUI_ACTIVE / 8 XCDs / time, MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)")
for k in sorted(cnt):
    g = cnt[k].get("GRBM_GUI_ACTIVE", [])
    m = cnt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
    if not g or not dur[k]:
        continue
    cyc = sum(g) / 8.0
    t_us = sum(dur[k])
    line = "%-16s n=%5d  mean %8.1f us  clock %.3f GHz" % (k, len(g), t_us / len(dur[k]), cyc / (t_us * 1e3))
    if m:
        line += "  MFMA busy %.3f" % (sum(m) / (1024.0 * cyc))
    print(line)
