import os, sys
sys.path.insert(0, '/root/repo')
import torch
from occuseg_amd import SCN, _lib, scenes
from tools.ab_tile6 import timeit
dev = torch.device("cuda:0")
coords, _ = scenes.batch([1000 + i for i in range(4)], 150000)
md = SCN.Metadata_3()
size = torch.tensor([4096] * 3)
md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
f3, f2 = torch.tensor([3, 3, 3]), torch.tensor([2, 2, 2])
rb0 = md.getSubmanifoldRuleBook(size, f3); n0 = md.getNActive(size)
md.getRuleBook(size, size // 2, f2, f2)
rb1 = md.getSubmanifoldRuleBook(size // 2, f3); n1 = md.getNActive(size // 2)
for (rb, n, c) in ((rb0, n0, 64), (rb1, n1, 128)):
    w = torch.randn(27, c, c, device=dev) * 0.05
    res = torch.randn(n, c, device=dev)
    flops = 2.0 * rb.total_pairs * c * c
    out = torch.empty(0, device=dev)
    for name, x in (("dense randn", torch.randn(n, c, device=dev)), ("relu(randn) (half zeros)", torch.relu(torch.randn(n, c, device=dev))),
                    ("tiny values 1e-30", torch.randn(n, c, device=dev) * 1e-30), ("denormals 1e-40", torch.randn(n, c, device=dev) * 1e-40)):
        line = "N=%d %d->%d input %-26s" % (n, c, c, name)
        for mode in ("fwd", "fwd+res", "fwd+stats", "fwd+res+stats", "dInput"):
            def f():
                if mode == "dInput":
                    SCN._conv(x, out, w, None, rb, 0, 1, n, True, True)
                else:
                    SCN._conv(x, out, w, None, rb, 0, 1, n, residual=res if "res" in mode else None, stats_out=[] if "stats" in mode else None)
            ms = timeit(f)
            line += "  %s %.1f" % (mode, flops / ms / 1e9)
        print(line, flush=True)
