#!/usr/bin/env python3
"""What the next batch's rule build costs the train step it runs beside: the bench's step (two alternating batches) with the
rulebooks (a) prefetched on the side stream (the default), (b) built serially in the forward pass, (c) not built at all -- the
two batches' Metadata built once and adopted again and again (an upper bound of what a cheaper rule build could return).
usage: python tools/rules_cost_in_step.py"""
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
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
batches = []
for base in (1000, 1100):
    coords, feats = scenes.batch([base + i for i in range(4)], 150000)
    batches.append([torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4])
input_layer = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))
counter = [0]
saved = {}


def step(mode):
    i = counter[0]
    counter[0] += 1
    b = batches[i % 2]
    if mode == "reuse":
        if i % 2 in saved:
            SCN._PREFETCHED[id(b[0])] = saved[i % 2]
    bucket.zero()
    net(b).sum().backward()
    bucket.all_reduce_mean()
    opt.step()
    nb = batches[(i + 1) % 2]
    if mode == "prefetch":
        input_layer.prefetch(nb, wait_for_current_stream=False)
    elif mode == "reuse" and (i + 1) % 2 not in saved:
        input_layer.prefetch(nb, wait_for_current_stream=False)
        saved[(i + 1) % 2] = SCN._PREFETCHED[id(nb[0])]
    elif mode == "serial":
        SCN._PREFETCHED.clear()


res = {}
for rnd in range(3):
    for mode in ("prefetch", "serial", "reuse"):
        SCN._PREFETCHED.clear()
        saved.clear()
        for _ in range(4):
            step(mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            step(mode)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10)
for mode, v in res.items():
    print("%-9s %s  min %.2f ms per step" % (mode, " ".join("%.2f" % t for t in v), min(v)))


# ---- which PART of the build costs the step what: the step adopts a saved Metadata (no build of its own) and a partial build runs
# beside it as pure load, where the prefetch would run (side stream, behind the backward pass' coarse-level mark)
def load_part(part, coords):
    side = SCN._SIDE_STREAMS.get((dev, SCN.PREFETCH_PRIORITY))
    if side is None:
        side = SCN._SIDE_STREAMS[(dev, SCN.PREFETCH_PRIORITY)] = torch.cuda.Stream(device=dev, priority=SCN.PREFETCH_PRIORITY)
    mark = SCN._COARSE_MARK.get(dev.index)
    if mark is not None and mark[1] == SCN._step_count(dev):
        side.wait_event(mark[0])
    size = torch.tensor([4096] * 3)
    f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
    with torch.cuda.stream(side):
        plans = dict(SCN._PLANS)
        SCN._PLANS.clear()  # (no replay: build exactly what is asked below)
        SCN.OCC_FIRST_BATCH = False  # (... piece by piece, by sort + hash)
        md = SCN.Metadata_3()
        md.inputLayer(size, coords, 0, 4, dev, _in_prefetch=True)  # grid of scale 0 (sort of the points)
        if part in ("grids", "all"):
            for lv in range(5):
                md.getRuleBook(size, size // 2, f2, f2)
                size = size // 2
        if part in ("sub0", "all"):
            size = torch.tensor([4096] * 3)
            for lv in range(6 if part == "all" else 1):
                md.getSubmanifoldRuleBook(size, f3)
                size = size // 2
        SCN._PLANS.clear()
        SCN._PLANS.update(plans)
    return md


res2 = {}
for rnd in range(2):
    for part in ("none", "grid0", "grids", "sub0", "all"):
        SCN._PREFETCHED.clear()
        saved.clear()
        keep = []
        times = []
        for it in range(14):
            if it == 4:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
            step("reuse")
            if part != "none":
                keep = [load_part(part, batches[counter[0] % 2][0])]
        e1.record()
        torch.cuda.synchronize()
        res2.setdefault(part, []).append(e0.elapsed_time(e1) / 10)
for part, v in res2.items():
    print("reuse + %-6s %s  min %.2f ms per step" % (part, " ".join("%.2f" % t for t in v), min(v)))
