#!/usr/bin/env python3
"""A/B of whole train steps with TWO builds of the library in ONE process (interleaved rounds, alternating two
different batches with the rulebook prefetch on, as bench.py does).  usage: python tools/ab_step_libs.py <lib1.so> [<lib2.so> ...]   (the product library is always the last contestant)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, _lib, dp, scenes

paths = [os.path.abspath(a) for a in sys.argv[1:]] + [_lib.LIB_PATH]
names = [chr(ord("A") + i) for i in range(len(paths))]
libs = {n: _lib._open(p, _lib.SIGNATURES) for n, p in zip(names, paths)}
_lib._lib = libs[names[-1]]
scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_net(scn, 64).to(dev)
bucket = dp.FlatGradBucket(net.parameters())
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
batches = []
for base in (1000, 1100):
    coords, feats = scenes.batch([base + i for i in range(4)], 150000)
    batches.append([torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4])
inp = next(m for m in net.modules() if isinstance(m, scn.InputLayer))
count = [0]


def step():
    i = count[0]
    count[0] += 1
    bucket.zero()
    net(batches[i % 2]).sum().backward()
    opt.step()
    inp.prefetch(batches[(i + 1) % 2], wait_for_current_stream=False)


for _ in range(4):
    step()
res = {n: [] for n in names}
for rnd in range(5):
    for name in names:
        torch.cuda.synchronize()
        _lib._lib = libs[name]
        SCN.invalidate_weight_cache()
        step()
        step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(6):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 6)
for name, path in zip(names, paths):
    print("%s (%s): %s  min %.2f ms" % (name, os.path.basename(path), " ".join("%.2f" % t for t in res[name]), min(res[name])))
