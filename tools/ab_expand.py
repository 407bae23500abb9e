import sys, os
sys.path.insert(0, os.getcwd())
import torch
from occuseg_amd import SCN, _lib, scenes
dev = torch.device("cuda:0")
lib = _lib.use_tuning_build()
coords, _ = scenes.batch([1000, 1001, 1002, 1003], 150000)
md = SCN.Metadata_3(); size = torch.tensor([4096] * 3)
md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
f2 = torch.tensor([2, 2, 2])
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for lv in range(3):
    n = md.getNActive(size)
    p = 64 * (lv + 1)
    rb = md.getRuleBook(size, size // 2, f2, f2)
    nc = md.getNActive(size // 2)
    torch.manual_seed(lv)
    # NiN 2p -> p over the identity list (SCN.NetworkInNetwork_updateOutput), deconv (p+64) -> p, strided-conv dInput
    x2 = torch.randn(n, 2 * p, device=dev); wn = torch.randn(2 * p, p, device=dev) * 0.05
    xc = torch.randn(nc, p + 64, device=dev); wd = torch.randn(8, p + 64, p, device=dev) * 0.05
    res = {}
    for mode in (0, 1, 0, 1):
        lib.scn_set_tuning(21, mode)
        o1 = torch.empty(0, device=dev)
        t1 = timeit(lambda: SCN.NetworkInNetwork_updateOutput(x2, o1, wn, torch.empty(0, device=dev)))
        o2 = torch.empty(0, device=dev)
        t2 = timeit(lambda: SCN.Deconvolution_updateOutput(size // 2, size, f2, f2, md, xc, o2, wd, torch.empty(0, device=dev)))
        key = (o1.clone(), o2.clone())
        same = "" if mode not in res else " bitwise=%s/%s" % (torch.equal(res[mode ^ 1][0], key[0]) if (mode ^ 1) in res else None, torch.equal(res[mode ^ 1][1], key[1]) if (mode ^ 1) in res else None)
        res[mode] = key
        print("L%d n=%d persist=%d  NiN %d->%d %.4f ms %.1f TF | deconv %d->%d (%d rows) %.4f ms %.1f TF%s" % (
            lv, n, mode, 2 * p, p, t1, 2.0 * n * 2 * p * p / t1 / 1e9, p + 64, p, n, t2, 2.0 * n * (p + 64) * p / t2 / 1e9, same), flush=True)
    size = size // 2
