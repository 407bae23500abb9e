#!/usr/bin/env python3
"""cfg2 (one 150k-voxel scene through the m32 UNet, forward only) under the switches that change its launch count.
usage: python tools/cfg2_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")


def timeit(fn, iters, warm):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
coords, feats = scenes.batch([1000], 150000)
net = bench.build_net(scn, 32).to(dev)
inp = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 1]
import time
with torch.no_grad():
    for rnd in range(3):
        for name, val in (("SPLIT_FUSED", True), ("SPLIT_FUSED", False)):
            setattr(SCN, name, val)
            ms = timeit(lambda: net(inp), 20, 5)
            # host time of one forward pass with the GPU idle at its start (how long issuing takes)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net(inp)
            host = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            print("%s=%s: %.3f ms per forward pass, host issue time %.3f ms" % (name, val, ms, host), flush=True)
SCN.SPLIT_FUSED = True
