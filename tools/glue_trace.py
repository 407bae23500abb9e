#!/usr/bin/env python3
"""Who launches the small torch kernels of a train step (fills, copies, cats): torch.profiler with Python stacks over two
steps, aggregated by (aten op, innermost frames of this repository).  Tuning tool."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import occuseg_amd
from occuseg_amd import SCN, dp, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_net(scn, 64).to(dev)
bucket = dp.FlatGradBucket(net.parameters())
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
batches = []
for s in (1000, 1100):
    coords, feats = scenes.batch([s, s + 1, s + 2, s + 3], 150000)
    batches.append([torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4])


def step(i):
    bucket.zero()
    out = net(batches[i % 2])
    SCN.prefetch_input(torch.tensor([4096] * 3), batches[(i + 1) % 2][0], 4, dev)
    out.sum().backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        step(i)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    if e.name not in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::mul",
                      "aten::sum", "aten::index", "aten::clone", "aten::contiguous", "aten::_foreach_add_"):
        continue
    dt = sum(k.duration for k in e.kernels) if e.kernels else 0.0
    if not e.kernels:
        continue
    frames = [f for f in (e.stack or []) if root in f and "glue_trace" not in f][:3]
    agg[(e.name, " <- ".join(f.replace(root + "/", "") for f in frames))][0] += 1
    agg[(e.name, " <- ".join(f.replace(root + "/", "") for f in frames))][1] += dt
tot = 0.0
for (name, where), (n, dt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%6.1f per step %8.1f us per step  %-16s %s" % (n / N, dt / N, name, where or "(no frame of this repository)"))
    tot += dt / N
print("total %.1f us per step" % tot)
