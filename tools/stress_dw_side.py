#!/usr/bin/env python3
"""Stress of the dWeight side stream: direct accumulation into a FlatGradBucket over two backward passes against the plain
autograd path, many times, in a few variants of the join.  usage: python tools/stress_dw_side.py [variant ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import occuseg_amd
from occuseg_amd import SCN, dp, scenes

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_unet import _net  # noqa: E402

scn = occuseg_amd.install()
gpu = torch.device("cuda:0")
torch.manual_seed(5)
net = _net(scn, 16, 3, 0.0).to(gpu)
coords, feats = scenes.batch([1000, 1001], target_sites=6000)
batch = [torch.from_numpy(coords).to(gpu), torch.from_numpy(feats).to(gpu), None, 2]
for p in net.parameters():
    p.grad = None
net(batch).sum().backward()
net(batch).sum().backward()
plain = [p.grad.clone() for p in net.parameters()]
names = [n for n, _ in net.named_parameters()]
bucket = dp.FlatGradBucket(net.parameters())
variants = sys.argv[1:] or ["side", "off", "hostsync_each", "hostsync_end"]
orig_end, orig_finish = SCN._dw_side_end, SCN._dw_finish
for v in variants:
    SCN.DW_SIDE_STREAM = v != "off"
    SCN._dw_side_end, SCN._dw_finish = orig_end, orig_finish
    if v == "hostsync_each":
        def end(side, d_out, _o=orig_end):
            _o(side, d_out)
            side.synchronize()
        SCN._dw_side_end = end
    if v == "hostsync_end":
        def fin(idx, _o=orig_finish):
            _o(idx)
            torch.cuda.synchronize()
        SCN._dw_finish = fin
    bad = 0
    worst = {}
    for it in range(20):
        bucket.zero()
        net(batch).sum().backward()
        net(batch).sum().backward()
        bucket.all_reduce_mean()
        for n, p, ref in zip(names, net.parameters(), plain):
            err = (p.grad - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
            if err > 1e-4:
                bad += 1
                worst[n] = max(worst.get(n, 0.0), err)
    print("%-14s mismatching (iteration, parameter) pairs: %d of %d   %s" % (v, bad, 20 * len(names), sorted(worst.items(), key=lambda kv: -kv[1])[:4]), flush=True)
bucket.close()
