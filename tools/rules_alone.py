#!/usr/bin/env python3
"""The rule build of one m64 batch (six grids, six submanifold and five strided rulebooks) alone on the GPU, ten times:
under `rocprofv3 --kernel-trace --stats` its per-kernel cost without the train step's kernels beside it.
usage: python tools/rules_alone.py [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, scenes


def build(coords, dev):
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, coords, 0, 4, dev)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    for lv in range(6):
        md.getSubmanifoldRuleBook(size, f3, _defer=True)
        if lv < 5:
            md.getRuleBook(size, size // 2, f2, f2, _defer=True)
            size = size // 2
    return md


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
    c = torch.from_numpy(coords).to(dev)
    build(c, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        build(c, dev)
    torch.cuda.synchronize()
    print("rule build alone: %.3f ms per batch (host clock, %d iterations)" % ((time.perf_counter() - t0) / iters * 1e3, iters))


if __name__ == "__main__":
    main()
