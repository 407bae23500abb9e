#!/usr/bin/env python3
"""dWeight launches of the train step's shapes, alone: float atomics / fixed order inside the kernel (workspace + dw_state) /
memset + kernel + reduction kernel (workspace only).  ms per call, events over 20 calls.  usage: python tools/ab_dw_det.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import occuseg_amd
from occuseg_amd import SCN, _lib, scenes
from occuseg_amd.SCN import check, ptr, stream

dev = torch.device("cuda:0")
lib = _lib.use_tuning_build() if os.environ.get("DW_BIG") else _lib.load()
coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
md = SCN.Metadata_3()
size = torch.tensor([4096] * 3)
md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
state = torch.zeros(_lib.DW_STATE_BYTES, dtype=torch.uint8, device=dev)
for lv, shapes in enumerate((((64, 64), (128, 64)), ((128, 128), (256, 128)), ((192, 192), (384, 192)), ((256, 256),))):
    rb = md.getSubmanifoldRuleBook(size, f3)
    n = md.getNActive(size)
    for ci, co in shapes:
        x, dy = torch.randn(n, ci, device=dev), torch.randn(n, co, device=dev)
        dw = torch.zeros(27, ci, co, device=dev)
        wsb = lib.scn_conv_dweight_workspace_bytes(27, rb.stride, n, n, ci, co, 1)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        line = "L%d N=%d %d->%d slots %.0f MB:" % (lv, n, ci, co, wsb / 1e6)
        modes = (("atomic", None, None), ("in-kernel", ws, state), ("two-launch", ws, None), ("atomic", None, None), ("in-kernel", ws, state))
        if os.environ.get("DW_BIG"):  # 128 x 128 blocks (tuning key 25) against 64 x 64, fixed order in-kernel and atomics
            big = int(os.environ["DW_BIG"])  # 1: 128 x 128 blocks, 2: 128 x 64
            modes = (("64 in-kernel", ws, state), ("128 in-kernel", ws, state), ("64 atomic", None, None), ("128 atomic", None, None)) * 2
        for name, w, st in modes:
            if os.environ.get("DW_BIG"):
                lib.scn_set_tuning(25, big if name.startswith("128") else 0)
            def call():
                check(lib.scn_conv_dweight(ptr(x), ptr(dy), ptr(dw), ptr(rb.rules), ptr(rb.counts), ptr(rb.bases.get(1)), 27, rb.stride,
                                           0, 1, n, n, ci, co, 0, 0, ptr(w), wsb if w is not None else 0, ptr(st), stream()))
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record()
            torch.cuda.synchronize()
            line += "  %s %.3f" % (name, e0.elapsed_time(e1) / 20)
        print(line, flush=True)
    if lv < 3:
        md.getRuleBook(size, size // 2, f2, f2)
        size = size // 2
