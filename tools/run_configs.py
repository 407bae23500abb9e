#!/usr/bin/env python3
"""Measure the non-headline BASELINE.json configs on one GPU: what bench.py puts into the driver's line as `configs`
(bench.other_configs): cfg2 = 1 scene ~150k voxels, m32 UNet, forward only; cfg5a / cfg5b = 2M-voxel single SC 128 -> 128."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import occuseg_amd

scn = occuseg_amd.install()
which = tuple(sys.argv[1:]) or ("cfg2", "cfg5a", "cfg5b")
print(json.dumps(bench.other_configs(scn, torch.device("cuda:0"), which), indent=1))
