#!/usr/bin/env python3
"""cProfile of the host side of cfg2's forward pass (m32 UNet, one 150k-voxel scene, no_grad): where the ~4.5 ms of issue
time go.  Tuning tool.  usage: python tools/cfg2_host.py [rows]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
coords, feats = scenes.batch([1000], 150000)
net = bench.build_net(scn, 32).to(dev)
inp = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 1]
with torch.no_grad():
    for _ in range(10):
        net(inp)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net(inp)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("host issue time per forward pass: min %.3f ms, median %.3f ms" % (min(ts), sorted(ts)[5]))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        net(inp)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
    st.sort_stats("cumulative").print_stats(30)
