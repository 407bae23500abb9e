#!/usr/bin/env python3
"""How much of a train step has NO kernel running: union of the kernel intervals of a rocprofv3 --kernel-trace against the
span of the timed steps.  usage: gpu_idle.py <dir of the trace>"""
import csv
import glob
import sys

iv = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
iv.sort()
# the steps: from the first to the last k_conv_tile4 launch
t4 = [i for i in iv if "k_conv_tile4" in i[2]]
lo, hi = t4[len(t4) // 3][0], t4[-1][1]  # skip the first third (warm-up)
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, _ in iv:
    if e < lo or s > hi:
        continue
    s, e = max(s, lo), min(e, hi)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = hi - lo
gaps.sort()
print("span %.2f ms, some kernel running %.2f ms (%.1f %%), idle %.2f ms in %d gaps (median %.1f us, largest %.1f us, gaps > 20 us: %d summing %.2f ms)"
      % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps), gaps[len(gaps) // 2] / 1e3 if gaps else 0,
         gaps[-1] / 1e3 if gaps else 0, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
