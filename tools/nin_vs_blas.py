import sys, torch
sys.path.insert(0, '/root/repo')
from occuseg_amd import SCN
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n, ci, co in ((600000, 128, 64), (186000, 256, 128), (44000, 384, 192), (7072, 512, 256)):
    x = torch.randn(n, ci, device=dev); w = torch.randn(ci, co, device=dev) * 0.05; g = torch.randn(n, co, device=dev)
    extra = torch.randn(n, ci, device=dev)
    out = torch.empty(0, device=dev); gi = torch.empty(0, device=dev); gw = torch.zeros(ci, co, device=dev)
    o2 = torch.empty(n, co, device=dev); gi2 = torch.empty(n, ci, device=dev)
    t_f = timeit(lambda: SCN.NetworkInNetwork_updateOutput(x, out, w, torch.empty(0, device=dev)))
    t_fm = timeit(lambda: torch.mm(x, w, out=o2))
    t_b = timeit(lambda: SCN.NetworkInNetwork_updateGradInput(gi, g, w))
    t_bm = timeit(lambda: torch.mm(g, w.t(), out=gi2))
    t_bma = timeit(lambda: torch.addmm(extra, g, w.t(), out=gi2))
    t_w = timeit(lambda: SCN.NetworkInNetwork_accGradParameters(x, g, gw, torch.empty(0, device=dev)))
    t_wm = timeit(lambda: torch.mm(x.t(), g, out=gw))
    byts = n * (ci + co) * 4
    print("n=%d %d->%d: fwd expand %.3f ms (%.2f TB/s) | mm %.3f ms (%.2f TB/s) || dIn expand %.3f | mm %.3f | addmm %.3f || dW tile %.3f | mm %.3f" % (
        n, ci, co, t_f, byts / t_f / 1e9, t_fm, byts / t_fm / 1e9, t_b, t_bm, t_bma, t_w, t_wm), flush=True)
