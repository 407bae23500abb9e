#!/usr/bin/env python3
"""Busy time and gaps of a `rocprofv3 --kernel-trace --output-format csv` trace of tools/rules_alone.py: splits the trace at
the first kernel of a batch (k_chain_count, else the key kernel of the int64 input coordinates), drops the first three
batches, and prints per batch the sum of the
kernel durations, the idle time between kernels, and the kernels by total time.
usage: python tools/rules_timeline.py <..._kernel_trace.csv>"""
import csv
import collections
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    head = "k_chain_count" if any("k_chain_count" in e[2] for e in ev) else "k_make_keys<1>"
    batches, cur = [], []
    for e in ev:
        if head in e[2] and cur:
            batches.append(cur)
            cur = []
        cur.append(e)
    batches = [b for b in batches if len(b) > 30][3:]  # (the last, open one is dropped)
    per = collections.defaultdict(lambda: [0, 0])
    span = busy = 0
    for b in batches:
        span += b[-1][1] - b[0][0]
        for s, e, n in b:
            busy += e - s
            k = n.split("(")[0][-60:]
            per[k][0] += 1
            per[k][1] += e - s
    nb = len(batches)
    print("%d batches: span %.1f us, kernels %.1f us, idle %.1f us, %d launches per batch" %
          (nb, span / nb / 1e3, busy / nb / 1e3, (span - busy) / nb / 1e3, sum(v[0] for v in per.values()) // nb))
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %-62s %5.1f launches %8.1f us per batch" % (k, v[0] / nb, v[1] / nb / 1e3))


if __name__ == "__main__":
    main()
