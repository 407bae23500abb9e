#!/usr/bin/env python3
"""Summarise the per-group rocprofv3 counter CSVs written by tools/pmc_groups.sh: per kernel-name pattern, the mean of
every counter per dispatch, plus the derived ratios DESIGN.md quotes.  usage: pmc_summary.py <outdir> [pattern]"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_conv_tile4"
vals = collections.defaultdict(list)
durs = []
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(out + "/g*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
m = {k: sum(v) / len(v) for k, v in vals.items()}
print("# kernel pattern %s: %d counters, %d dispatches per group" % (pat, len(m), len(next(iter(vals.values()))) if vals else 0))
if durs:
    m["avg_dispatch_us"] = sum(durs) / len(durs)
for k in sorted(m):
    print("%-36s %.4g" % (k, m[k]))
g = m.get
if g("GRBM_GUI_ACTIVE") and durs:
    cyc = g("GRBM_GUI_ACTIVE") / 8.0  # summed over the 8 XCDs
    print("clock_GHz %.3f" % (cyc / (m["avg_dispatch_us"] * 1e3)))
    if g("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("MFMA_pipe_busy %.3f  (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles))" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc)))
wc = g("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_VMEM_RD", "SQ_VMEM_TA_ADDR_FIFO_FULL",
              "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL"):
        if g(k) is not None:
            print("%-30s / SQ_WAVE_CYCLES = %.3f" % (k, g(k) / wc))
