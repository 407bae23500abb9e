#!/usr/bin/env python3
"""Per-launch timing of the gather-mm kernels inside one full train step (tuning tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, dp, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_net(scn, 64).to(dev)
bucket = dp.FlatGradBucket(net.parameters())
coords, feats = scenes.batch([1000, 1001, 1002, 1003], 150000)
batch_in = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4]


def step():
    bucket.zero()
    net(batch_in).sum().backward()


for _ in range(2):
    step()
SCN.PROFILE = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
step()
e1.record()
torch.cuda.synchronize()
prof, SCN.PROFILE = SCN.PROFILE, None
print("step (no optimizer) %.2f ms" % e0.elapsed_time(e1))
tot = {}
for kernel, flops, nbytes, a, b, shape in prof:
    ms = a.elapsed_time(b)
    tot[kernel] = tot.get(kernel, 0) + ms
    print("%-9s K=%2d rows=%7d %4d->%4d T=%d  %7.3f ms  %6.1f TF  %7.0f GB/s" % (
        (kernel,) + shape + (ms, flops / ms / 1e9, nbytes / ms / 1e6)))
print(tot)
