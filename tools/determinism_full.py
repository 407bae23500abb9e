#!/usr/bin/env python3
"""Two runs of three full-size train steps (m64 UNet, 4 scenes x ~150k voxels, fused Adam, dWeight on the side stream, stream-K
forward launches, prefetched rulebooks) from the same seed: first-step gradients and final parameters compared BITWISE.
usage: python tools/determinism_full.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, dp, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
batches = []
for s in (1000, 1100):
    coords, feats = scenes.batch([s, s + 1, s + 2, s + 3], 150000)
    batches.append([torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 4])


def run():
    torch.manual_seed(0)
    net = bench.build_net(scn, 64).to(dev)
    bucket = dp.FlatGradBucket(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    first = None
    for i in range(3):
        bucket.zero()
        out = net(batches[i % 2])
        SCN.prefetch_input(torch.tensor([4096] * 3), batches[(i + 1) % 2][0], 4, dev)
        out.sum().backward()
        bucket.all_reduce_mean()
        if first is None:
            first = bucket.flat.clone()
        opt.step()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    bucket.close()
    return first, params


assert SCN.DETERMINISTIC_DW
g1, p1 = run()
g2, p2 = run()
print("first-step gradients bitwise equal:", bool(torch.equal(g1, g2)), " parameters after 3 steps bitwise equal:", bool(torch.equal(p1, p2)),
      " (%d parameters; stream-K gave up: %s)" % (p1.numel(), SCN.stream_k_gave_up()))
SCN.DETERMINISTIC_DW = False
g3, _ = run()
d = (g1 - g3).abs().max().item() / g1.abs().max().item()
print("against float atomics: max |difference| / max |gradient| = %.2e, bitwise equal: %s" % (d, bool(torch.equal(g1, g3))))
