#!/usr/bin/env python3
"""cProfile of the host side of one train step (tuning tool)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import dp, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_net(scn, 64).to(dev)
bucket = dp.FlatGradBucket(net.parameters())
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
coords, feats = scenes.batch([1000, 1001, 1002, 1003], 150000)
batch_in = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4]


def step():
    bucket.zero()
    out = net(batch_in)
    out.sum().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per step %.1f ms, wall per step %.1f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
