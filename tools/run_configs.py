#!/usr/bin/env python3
"""Measure the non-headline BASELINE.json configs on one GPU (results table of BASELINE.md):
cfg2  1 scene ~150k voxels, m32 UNet, forward only;  cfg5a/5b  2M-voxel single SC 128->128 fwd+bwd."""
import json
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


res = {}
# cfg2
torch.manual_seed(0)
coords, feats = scenes.batch([1000], 150000)
net = bench.build_net(scn, 32).to(dev)
inp = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 1]
with torch.no_grad():
    ms = timeit(lambda: net(inp), 10, 3)
    md = SCN.Metadata_3()
    md.inputLayer(torch.tensor([4096] * 3), inp[0], 0, 4, dev)
    n0 = md.nActive[md.inputSpatialSize]
res["cfg2_m32_fwd_1scene"] = {"active_voxels": n0, "ms": ms, "voxels_per_s": n0 / ms * 1e3}
# the same as a serving loop would run it: the NEXT scene's voxel hash and rulebooks built on a side stream while this scene's
# kernels run (InputLayer.prefetch; here the next scene is the same tensor, built afresh every time)
input_layer = next(m for m in net.modules() if isinstance(m, scn.InputLayer))


def fwd_pipelined():
    out = net(inp)
    input_layer.prefetch(inp, wait_for_current_stream=False)
    return out


with torch.no_grad():
    ms_p = timeit(fwd_pipelined, 10, 3)
SCN._PREFETCHED.clear()
res["cfg2_m32_fwd_1scene_rules_prefetched"] = {"active_voxels": n0, "ms": ms_p, "voxels_per_s": n0 / ms_p * 1e3}

# cfg5
f3 = torch.tensor([3, 3, 3])
for name, c in (("cfg5a_bernoulli160", scenes.bernoulli_grid(160, 0.488, 7, offset=2)),
                ("cfg5b_solid126", scenes.solid_block(126, offset=3))):
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(c), 0, 4, dev)
    t_rules = timeit(lambda: (md.submanifoldRuleBooks.clear(), md.getSubmanifoldRuleBook(size, f3, _defer=True)), 5, 2)
    rb = md.getSubmanifoldRuleBook(size, f3)
    n = md.getNActive(size)
    R = rb.total_pairs
    x = torch.randn(n, 128, device=dev)
    w = torch.randn(27, 128, 128, device=dev) * 0.02
    dy = torch.randn(n, 128, device=dev)
    out, gi, gw = torch.empty(0, device=dev), torch.empty(0, device=dev), torch.zeros(27, 128, 128, device=dev)
    e = torch.empty(0, device=dev)
    t_f = timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n), 10, 3)
    t_di = timeit(lambda: SCN._conv(dy, gi, w, None, rb, 0, 1, n, True, True), 10, 3)
    lib = occuseg_amd._lib.load()
    t_dw = timeit(lambda: occuseg_amd._lib.check(lib.scn_conv_dweight(
        x.data_ptr(), dy.data_ptr(), gw.data_ptr(), rb.rules.data_ptr(), rb.counts.data_ptr(), None, 27, rb.stride, 0, 1,
        n, n, 128, 128, 0, 0, None, 0, None, occuseg_amd._lib.stream())), 10, 3)
    flops = 2.0 * R * 128 * 128
    fb = 4.0 * (R * 128 + n * 128) + 8.0 * R + 4.0 * 27 * 128 * 128
    wb = 4.0 * R * 256 + 8.0 * R + 4.0 * 27 * 128 * 128
    res[name] = {"active_voxels": n, "nu": R / n, "rulebook_build_ms": t_rules,
                 "fwd_ms": t_f, "fwd_TFLOPs": flops / t_f / 1e9, "fwd_algo_GBs": fb / t_f / 1e6,
                 "dinput_ms": t_di, "dinput_TFLOPs": flops / t_di / 1e9,
                 "dweight_ms": t_dw, "dweight_TFLOPs": flops / t_dw / 1e9, "dweight_algo_GBs": wb / t_dw / 1e6,
                 "fwd_bwd_voxels_per_s": n / (t_f + t_di + t_dw) * 1e3}
print(json.dumps(res, indent=1))
