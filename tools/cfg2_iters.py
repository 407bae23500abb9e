#!/usr/bin/env python3
"""cfg2 (one 150k-voxel scene, m32 UNet, forward only): wall time of every forward pass, one by one (allocator growth and other
first-iterations effects show as outliers), under tuning switches.  usage: python tools/cfg2_iters.py [tuneK=v ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd
from occuseg_amd import SCN, _lib, scenes

scn = occuseg_amd.install()
dev = torch.device("cuda:0")
torch.manual_seed(0)
coords, feats = scenes.batch([1000], 150000)
net = bench.build_net(scn, 32).to(dev)
inp = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 1]
sets = [a for a in sys.argv[1:]] or ["default"]
PREFETCH = os.environ.get("CFG2_PREFETCH", "0") == "1"  # the serving loop: the NEXT scene's rulebooks built on a side stream
input_layer = next(m for m in net.modules() if isinstance(m, scn.InputLayer))
with torch.no_grad():
    for rnd in range(2):
        for s in sets:
            if s != "default":
                for kv in s.split("+"):
                    k, v = kv.split("=")
                    if k.startswith("tune"):
                        _lib.use_tuning_build().scn_set_tuning(int(k[4:]), int(v))
                    else:
                        setattr(SCN, k, type(getattr(SCN, k))(int(v)) if not isinstance(getattr(SCN, k), str) else v)
            ts = []
            for it in range(16):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                net(inp)
                if PREFETCH:
                    input_layer.prefetch(inp, wait_for_current_stream=False)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3))
            print("%-28s wall ms: %s" % (s, " ".join("%.2f" % a for a, _ in ts)))
            print("%-28s host ms: %s" % ("", " ".join("%.2f" % b for _, b in ts)), flush=True)
