#!/usr/bin/env python3
"""The rule build of one m64 batch (six grids, six submanifold and five strided rulebooks) alone on the GPU: the
first call walks the scales as a UNet's forward pass does (that records the plan), every later InputLayer call replays
the plan -- what a training or inference loop does from its second batch on.  Under `rocprofv3 --kernel-trace` its
per-kernel cost and the gaps between the kernels (tools/rules_timeline.py) without a train step beside it.
usage: python tools/rules_alone.py [iterations]      (SCN_GRID_CHAIN=0|1)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, scenes


def first(coords, dev):
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, coords, 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    for lv in range(6):
        md.getSubmanifoldRuleBook(size, f3)
        if lv < 5:
            md.getRuleBook(size, size // 2, f2, f2)
            size = size // 2
    return md


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
    c = torch.from_numpy(coords).to(dev)
    first(c, dev)
    plan = list(SCN._PLANS_USED[(4096,) * 3])
    size = torch.tensor([4096] * 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(iters + 2):
        SCN._PLANS[(4096,) * 3] = list(plan)
        md = SCN.Metadata_3()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        md.inputLayer(size, c, 0, 4, dev)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ms.append((e0.elapsed_time(e1), (t1 - t0) * 1e3))
    ms = ms[2:]
    if os.environ.get("RULES_ALONE_ALL"):
        print("all (ms):", " ".join("%.2f" % m[0] for m in ms))
    ev = sorted(m[0] for m in ms)
    # (median: a generational collection of Python's garbage collector -- 35-40 ms in a process that has imported torch -- lands in
    #  one of some twenty iterations, wherever the allocation counts put it)
    print("rule build alone, GRID_CHAIN=%s OCC_BUILD=%s (built %d): %.3f ms per batch (events: median; min %.3f, max %.3f), host issue "
          "time %.3f ms (median), %d rulebooks, sites %s" % (
              SCN.GRID_CHAIN, SCN.OCC_BUILD, SCN.OCC_STATS["built"], ev[len(ev) // 2], ev[0], ev[-1],
              sorted(m[1] for m in ms)[len(ms) // 2], len(plan), [md.grids[k].n for k in sorted(md.grids, reverse=True)][:6]))


if __name__ == "__main__":
    main()
