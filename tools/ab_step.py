#!/usr/bin/env python3
"""A/B of whole train steps in ONE process (interleaved rounds): python tools/ab_step.py NAME=a,b ...
where NAME is a module-level switch of occuseg_amd.SCN (e.g. NIN_USE_GATHER_MM=1,0 PREBUILD=1,0)
or tuneK=v,w for scn_set_tuning(K, v)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, _lib, dp, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bench.build_net(scn, 64).to(dev)
bucket = dp.FlatGradBucket(net.parameters())
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
coords, feats = scenes.batch([1000, 1001, 1002, 1003], 150000)
batch_in = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4]


def step():
    bucket.zero()
    net(batch_in).sum().backward()
    opt.step()


def apply(name, val):
    if name.startswith("tune"):
        _lib.use_tuning_build().scn_set_tuning(int(name[4:]), int(val))
    elif "." in name:  # e.g. conv.EMIT_COLSTATS=1,0 : a switch of occuseg_amd.sparseconvnet.<module>
        import importlib
        mod, attr = name.split(".")
        m = importlib.import_module("occuseg_amd.sparseconvnet." + mod)
        setattr(m, attr, type(getattr(m, attr))(int(val)))
    else:
        cur = getattr(SCN, name)
        setattr(SCN, name, float(val) if isinstance(cur, float) else type(cur)(int(val)))


# AB_MAIN_PRIORITY=-1: run the steps on a high-priority stream (the dWeight side stream stays at normal priority)
_main = None
if os.environ.get("AB_MAIN_PRIORITY"):
    print("stream priority range", torch.cuda.Stream.priority_range())
    _main = torch.cuda.Stream(device=dev, priority=int(os.environ["AB_MAIN_PRIORITY"]))
    torch.cuda.set_stream(_main)
for fixed in os.environ.get("AB_FIXED", "").split():  # AB_FIXED="tune7=1 DW_SIDE_STREAM=0": settings held for the whole run
    apply(*fixed.split("="))
name, vals = sys.argv[1].split("=")
vals = vals.split(",")
for _ in range(3):
    step()
res = {v: [] for v in vals}
for rnd in range(4):
    for v in vals:
        apply(name, v)
        step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(4):
            step()
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 4)
for v in vals:
    print("%s=%s: %s  min %.2f ms" % (name, v, " ".join("%.2f" % t for t in res[v]), min(res[v])))
