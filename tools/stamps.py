#!/usr/bin/env python3
"""In-kernel stamp profile of the pipelined gather-mm kernel (diagnostic build, shares only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from occuseg_amd import SCN, _lib, scenes
dev = torch.device("cuda:0"); lib = _lib.use_tuning_build()
coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
md = SCN.Metadata_3(); size = torch.tensor([4096] * 3)
md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
names = ["store_rows(+vmcnt)", "barrier1", "issue loads", "group passes", "tail", "barrier2"]
if os.environ.get("STAMP_VARIANT") == "4":
    names = ["row offsets + weight scalars (pieces 0-2)", "weight loads + tile reads (3-7)", "2 index loads (8)", "8 row loads + scatter + table entry (9-15)"]
    lib.scn_set_tuning(0, 4)
for lv in range(2):
    rb = md.getSubmanifoldRuleBook(size, f3); n = md.getNActive(size); p = 64 * (lv + 1)
    for ci, co in ((p, p), (2 * p, p)):
        x = torch.randn(n, ci, device=dev); w = torch.randn(27, ci, co, device=dev) * .05; out = torch.empty(0, device=dev)
        lib.scn_set_tuning(1, int(os.environ.get('STAMP_DBG', '256')))
        buf = (ctypes.c_ulonglong * 8)()
        SCN._conv(x, out, w, None, rb, 0, 1, n); torch.cuda.synchronize(); lib.scn_debug_stamps(buf)
        SCN._conv(x, out, w, None, rb, 0, 1, n); torch.cuda.synchronize(); lib.scn_debug_stamps(buf)
        lib.scn_set_tuning(1, 0)
        if os.environ.get("STAMP_VARIANT") == "4":
            np_ = buf[4]
            print("L%d %d->%d: cycles per pass in the pieces between MFMAs: " % (lv, ci, co) + "  ".join("%s %.0f" % (nm, buf[i] / np_) for i, nm in enumerate(names)))
            continue
        tot = sum(buf[:6])
        print("L%d %d->%d: " % (lv, ci, co) + "  ".join("%s %.1f%%" % (nm, 100.0 * buf[i] / tot) for i, nm in enumerate(names)))
    md.getRuleBook(size, size // 2, f2, f2); size = size // 2
