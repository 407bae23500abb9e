#!/usr/bin/env python3
"""bench.py -- active-voxels/s of one fwd+bwd train step of the ScanNet m64 UNet backbone.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: InputLayer (voxel hash + every rulebook is
rebuilt each step, as in real training) -> stem SC(3->m) -> scn.UNet(3,1,[m..6m],residual) ->
BatchNormReLU -> OutputLayer -> sum loss -> backward -> (N>1: RCCL all-reduce of the flat
gradient) -> Adam step.  Workload = BASELINE.json configs[2]: 4 synthetic ScanNet-like scenes of
~150k active voxels per GPU (weak scaling: configs[3] is 8 x this).  Every rank alternates between TWO
different resident batches (different scenes, different site counts), so no step sees the coordinates of the
step before it; while a step's kernels drain, the OTHER batch's voxel hash and rulebooks are built on a side
stream (InputLayer.prefetch), as a data loader would do.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import gc
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")


def build_net(scn, m):
    return scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
        scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, [m, 2 * m, 3 * m, 4 * m, 5 * m, 6 * m], residual_blocks=True)).add(
        scn.BatchNormReLU(m)).add(scn.OutputLayer(3))


def source_hash():
    """identity of the kernel sources a measurement belongs to (there is no .git on the GPU box)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "occuseg_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """cores this process may really use: min(affinity mask, cgroup v2 cpu.max quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def device_pass(net, scn, batch):
    """One forward + backward (loss = sum of the outputs) of `net` through the HIP path on `batch`, kept for the comparison with
    the reference's compiled CPU loops (parity_fullsize): output, every parameter gradient in module order, and the sign of
    every batch norm's output as the device computed it (the CPU backward pass then differentiates the same piecewise-linear
    function: tests/helpers.py device_masks).  Product code only; nothing here touches oracle/."""
    masks, hooks = [], []
    for mod in net.modules():
        if isinstance(mod, scn.BatchNormalization):
            hooks.append(mod.register_forward_hook(lambda md, inp, out: masks.append((out.features.detach() > 0).cpu().numpy())))
    for p in net.parameters():
        p.grad = None
    try:
        out = net(batch)
        out.sum().backward()
    finally:
        for h in hooks:
            h.remove()
    torch.cuda.synchronize()
    grads = []
    for mod in net.modules():
        for name, p in mod._parameters.items():
            if p is not None:
                grads.append(("%s.%s" % (type(mod).__name__, name), tuple(p.shape), p.grad.detach().cpu().numpy()))
    res = {"out": out.detach().cpu().numpy(), "grads": grads, "masks": masks}
    for p in net.parameters():
        p.grad = None
    return res


def compare_passes(dev_pass, cpu_out, cpu_grads_in_order, mask_flips, mask_total):
    """parity_fullsize: the HIP pass against the reference's compiled loops on the same batch and parameters.  `*_max_rel`
    = max |got - want| / max |want| per tensor (the bar of north_star: 1e-4; tests/helpers.py assert_close), the worst
    tensor named; `elementwise_rel_p99` / `_p999`: quantiles of |got - want| / |want| over the output elements with
    |want| >= 1e-3 max |want| (elements that cancel to ~0 cannot be held to a per-element ratio)."""
    got, want = dev_pass["out"].astype(np.float64), np.asarray(cpu_out, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    diff = np.abs(got - want)
    sel = np.abs(want) >= 1e-3 * scale
    ew = np.sort((diff[sel] / np.abs(want[sel])).ravel())
    res = {"forward_max_rel": float(diff.max()) / scale, "forward_shape": list(got.shape),
           "elementwise_rel_p50": float(ew[len(ew) // 2]) if len(ew) else None,
           "elementwise_rel_p99": float(ew[int(0.99 * (len(ew) - 1))]) if len(ew) else None,
           "elementwise_rel_p999": float(ew[int(0.999 * (len(ew) - 1))]) if len(ew) else None,
           "elementwise_rel_max": float(ew[-1]) if len(ew) else None,
           "elementwise_over": "output elements with |want| >= 1e-3 max|want| (%d of %d)" % (int(sel.sum()), int(sel.size))}
    worst, worst_name, per = 0.0, None, []
    assert len(dev_pass["grads"]) == len(cpu_grads_in_order), (len(dev_pass["grads"]), len(cpu_grads_in_order))
    for (name, shape, g), (cname, cg) in zip(dev_pass["grads"], cpu_grads_in_order):
        assert name == cname, (name, cname)
        cg = np.asarray(cg, np.float64).reshape(shape)
        sc = max(float(np.abs(cg).max()), 1e-30)
        e = float(np.abs(g.astype(np.float64) - cg).max()) / sc
        per.append(e)
        if e >= worst:
            worst, worst_name = e, "%s %s (tensor %d of %d)" % (name, list(shape), len(per), len(cpu_grads_in_order))
    per = np.sort(np.array(per))
    res.update({"grad_max_rel": worst, "worst_tensor": worst_name, "grad_tensors": int(len(per)),
                "grad_median_rel": float(per[len(per) // 2]),
                "grad_masks": "the CPU backward pass ran over the device's activation masks (sign of every batch norm's output as the "
                              "HIP forward pass computed it): %d of %d mask elements differ from the CPU forward pass' own"
                              % (mask_flips, mask_total),
                "mask_flips": int(mask_flips), "mask_elements": int(mask_total),
                "tolerance": 1e-4, "ok": bool(res["forward_max_rel"] <= 1e-4 and worst <= 1e-4)})
    return res


def _cpu_once(scn, m, target_sites, seed):
    from occuseg_amd import scenes
    from oracle import torch_ref
    torch.manual_seed(0)
    coords, feats = scenes.batch([seed], target_sites)
    net = build_net(scn, m)
    t0 = time.perf_counter()
    out, stats = torch_ref.ref_forward(net, coords, feats, torch.float32)
    out.sum().backward()
    dt = time.perf_counter() - t0 - stats["ctx"].rule_seconds
    return stats["ctx"].grid0.n, dt


def cpu_baseline_port(m, target_sites, seed, budget_s=20.0):
    """Fallback when oracle/_ref/libscn_ref.so is absent: the oracle's torch port (numpy rulebooks + torch CPU
    index_select/mm/index_add_, float32, all host cores) on ONE scene sized to ~budget_s seconds."""
    import occuseg_amd
    scn = occuseg_amd.install()
    cores = host_cores()
    torch.set_num_threads(cores)
    n0, dt = _cpu_once(scn, m, 10000, seed)
    want = int(min(target_sites, max(10000, budget_s * n0 / dt)))
    if want > 1.5 * n0:
        n0, dt = _cpu_once(scn, m, want, seed)
    return {"value": n0 / dt, "unit": "active-voxels/s", "cores": cores, "kind": "port",
            "sample": "1 scene (seed %d) of %d active voxels, m%d UNet, one fwd+bwd in %.1f s, float32 torch CPU "
                      "mm + index_select/index_add_, rulebook build excluded" % (seed, n0, m, dt)}


def cpu_baseline(m, scenes_per_gpu, target_sites, seed0, iters=1, dev_pass=None):
    """the reference's OWN compiled CPU loops (oracle/_ref/libscn_ref.so = sparseconvnet/SCN/CPU/
    {Convolution,BatchNormalization,IOLayers}.cpp built from where they lie; oracle/ref_net.py walks the network) on the
    host cores of this box: one fwd+bwd of the same network over the SAME batch the GPU is timed on (4 scenes, seeds
    1000..) with every usable core, and over ONE of its scenes with a single thread (the full batch at one thread
    would take minutes).  Rulebooks come from the oracle and their construction is outside the clock."""
    from oracle import ref as REF
    if not REF.available():
        return cpu_baseline_port(m, target_sites, seed0)
    import occuseg_amd
    from occuseg_amd import scenes
    from oracle import ref_net
    scn = occuseg_amd.install()
    cores = host_cores()
    torch.manual_seed(0)
    net = build_net(scn, m)
    seeds = [seed0 + i for i in range(scenes_per_gpu)]
    coords, feats = scenes.batch(seeds, target_sites)
    ctx, best, groups = None, None, None
    # parity_fullsize (dev_pass = the HIP path's output and gradients on this very batch with these very parameters, taken
    # before the warm-up): the same forward + backward the clock is on, run over the device's activation masks
    masks = None
    if dev_pass is not None:
        bns = [mod for mod in net.modules() if isinstance(mod, scn.BatchNormalization)]
        assert len(bns) == len(dev_pass["masks"])
        masks = {id(mod): mk for mod, mk in zip(bns, dev_pass["masks"])}
    parity = None

    def in_order(g):
        return [("%s.%s" % (type(mod).__name__, name), g[(id(mod), name)])
                for mod in net.modules() for name, p in mod._parameters.items() if p is not None]

    for it in range(max(1, iters)):
        _out, _g, sec, ctx = ref_net.forward_backward(net, coords, feats, threads=cores, ctx=ctx, masks=masks)
        if best is None or sec["total"] < best:
            best, groups = sec["total"], sec
        if dev_pass is not None and it == 0:
            # (1) against the arithmetic the reference RUNS: its float32 loops, the pass the clock is on
            order32 = in_order(_g)
            p32 = compare_passes(dev_pass, _out, order32, ctx.mask_flips, ctx.mask_total)
            # (2) against the SAME templates of the reference instantiated at double (oracle/ref_harness.cpp *_f64): at this size
            # the float32 loops' own sequential sums over 600k rows (BatchNormalization.cpp:24-36,75-87) are ~1e-3 from that, so
            # the 1e-4 bar of north_star can only be read against the float64 instantiation; (3) says how far (1)'s oracle is from it
            t64 = time.perf_counter()
            out64, g64, _s64, ctx = ref_net.forward_backward(net, coords, feats, threads=cores, ctx=ctx, masks=masks, dtype=np.float64)
            order64 = in_order(g64)
            p64 = compare_passes(dev_pass, out64, order64, ctx.mask_flips, ctx.mask_total)
            ref32 = {"out": _out, "grads": [(n, tuple(np.shape(g)), np.asarray(g)) for n, g in order32]}
            r32 = compare_passes(ref32, out64, order64, 0, 0)
            keys = ("forward_max_rel", "grad_max_rel", "grad_median_rel", "worst_tensor", "elementwise_rel_p50", "elementwise_rel_p99",
                    "elementwise_rel_p999", "elementwise_rel_max")
            parity = dict(p64)
            parity["against"] = ("oracle/_ref: the reference's CPU loops (sparseconvnet/SCN/CPU/{Convolution,BatchNormalization,IOLayers}.cpp) "
                                 "instantiated at double, same batch (%d scenes, %d active voxels), same seed-0 parameters, the device's "
                                 "activation masks" % (len(seeds), ctx.grid0.n))
            parity["hip_vs_reference_f32_loops"] = {k: p32[k] for k in keys}
            parity["reference_f32_loops_vs_their_f64_instantiation"] = {k: r32[k] for k in keys}
            parity["note"] = ("top level = HIP against the float64 instantiation (the bar: 1e-4); the reference's own float32 loops are "
                              "themselves further from it than the HIP path is (sequential float32 sums over %d rows)" % ctx.grid0.n)
            parity["f64_leg_seconds"] = time.perf_counter() - t64
            del out64, g64, order64, ref32
        del _out, _g
    n_all = ctx.grid0.n
    del ctx, masks
    c1, f1 = scenes.batch(seeds[:1], target_sites)
    _out, _g, sec1, ctx1 = ref_net.forward_backward(net, c1, f1, threads=1)
    n_one = ctx1.grid0.n
    return {"value": n_all / best, "unit": "active-voxels/s", "cores": cores, "kind": "reference", "parity_fullsize": parity,
            "one_thread": {"value": n_one / sec1["total"], "cores": 1, "seconds": sec1["total"], "active_voxels": n_one,
                           "sample": "scene %d of the batch" % seeds[0]},
            "seconds": best, "seconds_by_group": groups, "active_voxels": n_all, "iterations": max(1, iters),
            "sample": "the full cfg3 batch: %d scenes (seeds %d..%d), %d active voxels, m%d UNet, one fwd+bwd in %.1f s "
                      "on %d threads through the reference's compiled CPU loops (rule_index_select -> at::mm -> "
                      "rule_index_add_ per offset, BatchNormalization_Forward/BackwardPass, InputLayer passes; "
                      "oracle/_ref); one scene at 1 thread beside it; rulebook build excluded"
                      % (len(seeds), seeds[0], seeds[-1], n_all, m, best, cores)}


def _timeit(fn, iters, warm):
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


def other_configs(scn, dev, which=("cfg2", "cfg5a", "cfg5b")):
    """BASELINE.json configs[1] and configs[4] on this GPU, after the timed region (N = 1 only, a few seconds): cfg2 = ONE
    ~150k-voxel scene through the m32 UNet, forward only (train-mode batch statistics), alone and with the next scene's rule
    build prefetched as a serving loop would; cfg5a / cfg5b = ONE SubmanifoldConvolution 128 -> 128 over 2 M active voxels
    (5a: Bernoulli 160^3, nu ~ 13.7; 5b: solid 126^3, nu ~ 26.6): forward, dInput, dWeight and the rulebook build, in TFLOP/s
    (2 R C_i C_o) and algorithmic GB/s (SURVEY.md section 8d's formulas).  tools/run_configs.py prints the same dict."""
    from occuseg_amd import SCN, _lib, scenes
    res = {}
    if "cfg2" in which:
        torch.manual_seed(0)
        coords, feats = scenes.batch([1000], 150000)
        net = build_net(scn, 32).to(dev)
        inp = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 1]
        input_layer = next(m for m in net.modules() if isinstance(m, scn.InputLayer))
        SCN._PREFETCHED.clear()
        with torch.no_grad():
            ms = _timeit(lambda: net(inp), 10, 3)
            md = SCN.Metadata_3()
            md.inputLayer(torch.tensor([4096] * 3), inp[0], 0, 4, dev)
            n0 = md.nActive[md.inputSpatialSize]
            del md

            def fwd_pipelined():  # the NEXT scene's voxel hash + rulebooks built on the side stream beside this scene's kernels
                out = net(inp)
                input_layer.prefetch(inp, wait_for_current_stream=False)
                return out
            ms_p = _timeit(fwd_pipelined, 10, 3)
        SCN._PREFETCHED.clear()
        res["cfg2_m32_fwd_1scene"] = {"active_voxels": n0, "ms": ms, "voxels_per_s": n0 / ms * 1e3,
                                      "ms_rules_prefetched": ms_p, "voxels_per_s_rules_prefetched": n0 / ms_p * 1e3}
        del net, inp
    f3 = torch.tensor([3, 3, 3])
    for name, key, make in (("cfg5a_bernoulli160", "cfg5a", lambda: scenes.bernoulli_grid(160, 0.488, 7, offset=2)),
                            ("cfg5b_solid126", "cfg5b", lambda: scenes.solid_block(126, offset=3))):
        if key not in which:
            continue
        c = make()
        md = SCN.Metadata_3()
        size = torch.tensor([4096] * 3)
        md.inputLayer(size, torch.from_numpy(c), 0, 4, dev)
        t_rules = _timeit(lambda: (md.submanifoldRuleBooks.clear(), md.getSubmanifoldRuleBook(size, f3, _defer=True)), 3, 1)
        rb = md.getSubmanifoldRuleBook(size, f3)
        n = md.getNActive(size)
        R = rb.total_pairs
        # ... and the whole batch as an InputLayer call builds it from its second batch on (the plan the call above recorded: this
        # scale's grid + its rulebook; bounds, occupancy bitmaps, sites, neighbours, lists, tile bases and both read-backs)
        cdev = torch.from_numpy(c).to(dev)
        plan = list(SCN._PLANS[(4096,) * 3])

        def batch_build():
            SCN._PLANS[(4096,) * 3] = list(plan)  # (what a forward pass would have recorded again)
            SCN.Metadata_3().inputLayer(size, cdev, 0, 4, dev)
        t_batch = _timeit(batch_build, 3, 1)
        x = torch.randn(n, 128, device=dev)
        w = torch.randn(27, 128, 128, device=dev) * 0.02
        dy = torch.randn(n, 128, device=dev)
        out, gi, gw = torch.empty(0, device=dev), torch.empty(0, device=dev), torch.zeros(27, 128, 128, device=dev)
        t_f = _timeit(lambda: SCN._conv(x, out, w, None, rb, 0, 1, n), 6, 2)
        t_di = _timeit(lambda: SCN._conv(dy, gi, w, None, rb, 0, 1, n, True, True), 6, 2)
        lib = _lib.load()
        t_dw = _timeit(lambda: _lib.check(lib.scn_conv_dweight(
            x.data_ptr(), dy.data_ptr(), gw.data_ptr(), rb.rules.data_ptr(), rb.counts.data_ptr(), None, 27, rb.stride, 0, 1,
            n, n, 128, 128, 0, 0, None, 0, None, _lib.stream())), 6, 2)
        flops = 2.0 * R * 128 * 128
        fb = 4.0 * (R * 128 + n * 128) + 8.0 * R + 4.0 * 27 * 128 * 128
        wb = 4.0 * R * 256 + 8.0 * R + 4.0 * 27 * 128 * 128
        res[name] = {"active_voxels": n, "nu": R / n, "rulebook_build_ms": t_rules, "grid_and_rulebook_batch_build_ms": t_batch,
                     "fwd_ms": t_f, "fwd_TFLOPs": flops / t_f / 1e9, "fwd_mfma_frac": flops / t_f / 1e9 / F32_MFMA_PEAK_TFLOPS,
                     "fwd_algo_GBs": fb / t_f / 1e6,
                     "dinput_ms": t_di, "dinput_TFLOPs": flops / t_di / 1e9,
                     "dweight_ms": t_dw, "dweight_TFLOPs": flops / t_dw / 1e9, "dweight_algo_GBs": wb / t_dw / 1e6,
                     "fwd_bwd_voxels_per_s": n / (t_f + t_di + t_dw) * 1e3}
        del x, w, dy, out, gi, gw, rb, md
        torch.cuda.empty_cache()
    return res


class _HostBatches(object):
    """a data loader's worth of behaviour for PrefetchingLoader: n batches cycling over pinned HOST copies of the resident ones"""

    def __init__(self, host_batches, first, n):
        self.host, self.first, self.n = host_batches, first, n

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            b = self.host[(self.first + i) % len(self.host)]
            yield {"x": [b[0], b[1], None, b[3]]}


def measured_peaks(dev):
    """f32-MFMA and HBM-stream microbenchmarks on THIS box (libscn_hip_tuning.so, outside the timed region)"""
    from occuseg_amd import _lib
    try:
        lib = _lib._open(_lib.TUNING_LIB_PATH, _lib.TUNING_SIGNATURES)
    except (RuntimeError, OSError, AttributeError):
        return None
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.zeros(64, device=dev)
    tf = ctypes.c_double(0)
    best_tf = 0.0
    for wps in (1, 2, 4):
        _lib.check(lib.scn_peak_mfma_f32(scratch.data_ptr(), wps, ctypes.byref(tf), stream))
        best_tf = max(best_tf, tf.value)
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).zero_()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    gbs = ctypes.c_double(0)
    _lib.check(lib.scn_peak_copy(src.data_ptr(), dst.data_ptr(), nbytes, 3, ctypes.byref(gbs), stream))
    rd = ctypes.c_double(0)
    _lib.check(lib.scn_peak_read(src.data_ptr(), scratch.data_ptr(), nbytes, 3, ctypes.byref(rd), stream))
    del src, dst
    return {"f32_mfma_tflops": best_tf, "hbm_copy_GBs": gbs.value, "hbm_read_GBs": rd.value,
            "how": "v_mfma_f32_16x16x4_f32 chains on every SIMD (1, 2 and 4 waves per SIMD, best); float4 copy of 1 GiB "
                   "(bytes read + written) and float4 read-only stream of 1 GiB, best of 3"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--m", "--planes", dest="m", type=int, default=64)
    ap.add_argument("--scenes-per-gpu", type=int, default=4)
    ap.add_argument("--sites", type=int, default=150000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip parity_fullsize (the HIP pass kept for the comparison with the CPU leg)")
    ap.add_argument("--no-configs", action="store_true", help="skip the cfg2 / cfg5 measurements behind the timed region")
    ap.add_argument("--host-steps", type=int, default=10,
                    help="steps of the second, from-host timed region (inputs pinned on the host, H2D inside the clock); 0 = skip")
    ap.add_argument("--cpu-iters", type=int, default=1, help="iterations of the all-cores CPU baseline (best is reported)")
    ap.add_argument("--event-every", type=int, default=10,
                    help="time the kernels with HIP events on every n-th step of the timed region (a timed event drains "
                         "the queue between two kernels: ~1200 of them cost a profiled step ~25 ms)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="build each step's rulebooks inside the step instead of one step ahead on a side stream")
    ap.add_argument("--no-dw-side-stream", action="store_true",
                    help="keep the dWeight kernels on the main stream in every step (kernel durations in a profile of this "
                         "command are then exclusive ones, as on the steps whose launches carry events)")
    ap.add_argument("--atomic-dweight", action="store_true",
                    help="SCN.DETERMINISTIC_DW = False: dWeight partial sums added with float atomics (the reference's way)")
    ap.add_argument("--deterministic", action="store_true",
                    help="SCN.DETERMINISTIC_DW: dWeight partial sums added in a fixed order (bitwise reproducible gradients)")
    ap.add_argument("--same-batch", action="store_true", help="replay ONE batch (round-1 behaviour) instead of alternating two")
    ap.add_argument("--dump-launches", default=None, help="write per-launch timings (kernel, shape, ms) to this JSON file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (how the driver calls N = 1): start one rank per GPU as CHILD processes through
        # torch.distributed.run and hand on its exit status.  Nothing in this process has touched the GPU (importing
        # torch does not), and nothing is exec'ed: the launcher is a child, rank 0's JSON line goes to the inherited stdout.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus))))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("SCN_BENCH_BACKEND", "nccl")  # tests: gloo (every rank on cuda:0 with SCN_BENCH_ONE_DEVICE=1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", 0 if os.environ.get("SCN_BENCH_ONE_DEVICE") else local_rank)
    torch.cuda.set_device(dev)

    import occuseg_amd
    from occuseg_amd import SCN, dp, scenes
    scn = occuseg_amd.install()
    if args.deterministic:
        SCN.DETERMINISTIC_DW = True
    if args.atomic_dweight:
        SCN.DETERMINISTIC_DW = False
    for sw in filter(None, os.environ.get("SCN_BENCH_SWITCHES", "").split(",")):  # debugging: e.g. conv.JOIN_IN_PLACE=0,tune13=0
        name, val = sw.split("=")
        if name.startswith("tune"):
            from occuseg_amd import _lib as _l
            _l.use_tuning_build().scn_set_tuning(int(name[4:]), int(val))
        elif "." in name:
            import importlib
            mod, attr = name.split(".")
            setattr(importlib.import_module("occuseg_amd.sparseconvnet." + mod), attr, int(val))
        else:
            setattr(SCN, name, int(val))

    torch.manual_seed(0)  # identical replicas
    net = build_net(scn, args.m).to(dev)
    n_params = sum(p.numel() for p in net.parameters())
    bucket = dp.FlatGradBucket(net.parameters())
    if world > 1:
        bucket.enable_overlap()  # bucketed all-reduce launched from backward as each bucket completes
        bucket.record_exposed = True
    # torch's fused Adam kernel (same arithmetic as the default foreach implementation: 53.5 -> 51.9 ms per step)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=bool(int(os.environ.get("SCN_BENCH_FUSED_ADAM", "1"))))

    # two resident batches per rank: scenes 1000.. and 1100.. (different geometry, different site counts)
    batches, n_sites = [], []
    for base in ((1000,) if args.same_batch else (1000, 1100)):
        all_seeds = [base + i for i in range(args.scenes_per_gpu * world)]
        seeds = dp.shard(all_seeds, rank, world)
        coords, feats = scenes.batch(seeds, args.sites)
        b = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, len(seeds)]
        with torch.no_grad():
            md = SCN.Metadata_3()
            md.inputLayer(torch.tensor([4096] * 3), b[0], 0, 4, dev)
            n_sites.append(md.nActive[md.inputSpatialSize])
            del md
        batches.append(b)
    input_layer = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))
    counter = [0]

    trace = (lambda *a: print("[bench]", *a, file=sys.stderr, flush=True)) if os.environ.get("SCN_BENCH_TRACE") else (lambda *a: None)

    def step():
        i = counter[0]
        counter[0] += 1
        trace("step", i)
        bucket.zero()
        out = net(batches[i % len(batches)])
        trace("forward issued")
        out.sum().backward()
        trace("backward issued")
        bucket.all_reduce_mean()
        opt.step()
        trace("optimizer issued")
        if args.prefetch:
            # software pipelining across steps: the NEXT batch's voxel hash + rulebooks are built on a side stream
            # while this step's kernels drain (every step still builds its own)
            input_layer.prefetch(batches[(i + 1) % len(batches)], wait_for_current_stream=False)
            trace("prefetch done")
        return out

    if args.no_dw_side_stream:  # (the warm-up steps too)
        SCN.DW_SIDE_STREAM = False
    # parity_fullsize: ONE forward + backward of the HIP path on the batch the CPU leg will run (scenes 1000..), with the
    # seed-0 parameters both sides start from, kept on the host until the reference's compiled loops have run (after the timed
    # region).  Nothing of oracle/ is imported or loaded here.
    dev_pass = None
    want_parity = (world == 1 and not args.no_cpu_baseline and not args.no_parity and args.m == 64
                   and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libscn_ref.so")))
    if want_parity:
        dev_pass = device_pass(net, scn, batches[0])
        SCN._PREFETCHED.clear()
    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every MATRIX kernel (gather-mm, dWeight), inside the timed region, on every EVENT_EVERY-th
    # step: a timed event drains the queue between two kernels (with events on ALL ~600 launches of a step -- batch
    # norm, rule build, io layers too -- a profiled step took 127 ms instead of 50).  The HBM-bound launches are
    # therefore timed in ONE extra step after the timed region.  One more event per step (end of step, main stream)
    # gives the per-step durations for the median.
    prof, prof_extra = [], []
    dw_side = bool(SCN.DW_SIDE_STREAM) and not args.no_dw_side_stream
    SCN.DW_SIDE_STREAM = dw_side
    SCN.PROFILE_KINDS = ("mm",)
    SCN.PROFILE_FUSED.clear()
    if rank == 0:  # ~230 events per profiled step (matrix launches) + ~900 for the extra step, created now
        SCN.warm_event_pool(260 * len(range(0, args.steps, args.event_every)) + 1000)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sites_done = 0
    first = counter[0]
    # Python's generational garbage collector walks every container object of the process when its oldest generation comes due: 35-40
    # ms in a process that has imported torch (profiles/r06_rules_alone.txt: one iteration in twenty of a 0.45 ms loop), at a moment the
    # allocation counts pick.  What a long-running training or serving loop does after its warm-up: collect once, then move what is
    # alive out of the collector's sight -- the timed steps allocate and free as before, nothing of the step is skipped.
    gc.collect()
    gc.freeze()
    barrier()
    wpack_before = dict(SCN.WPACK_STATS)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        SCN.PROFILE = prof if (i % args.event_every == 0 and rank == 0) else None
        # dWeight normally runs on a side stream beside the kernels that follow it: two kernels
        # sharing the chip both take longer, so on the steps whose launches carry events it stays on the main stream and the
        # per-kernel durations are EXCLUSIVE ones (every rank does the same: the steps stay aligned)
        SCN.DW_SIDE_STREAM = dw_side and (i % args.event_every != 0)
        step()
        marks[i + 1].record()
        sites_done += n_sites[(first + i) % len(batches)]
    SCN.PROFILE = None
    SCN.DW_SIDE_STREAM = False  # (the extra step below measures exclusive durations too)
    barrier()
    dt = time.perf_counter() - t0
    wpack_after = dict(SCN.WPACK_STATS)
    # the extra, untimed step (every rank runs it: it holds collectives): events on batch norm, rule build, io launches
    SCN.PROFILE_KINDS = ("bn", "rules", "io")
    SCN.PROFILE = prof_extra if rank == 0 else None
    seen_md = []
    hook = input_layer.register_forward_hook(lambda mod, inp, outp: seen_md.append(outp.metadata))
    step()
    hook.remove()
    SCN.PROFILE = None
    SCN.PROFILE_KINDS = ("mm", "bn", "rules", "io")
    barrier()

    # ---- a second, shorter timed region that starts where a training loop starts: the batches PINNED ON THE HOST
    # (occuseg_amd.data.PrefetchingLoader: coordinates + features cross PCIe on a copy stream while the step before drains, the
    # rule build on the side stream waits for exactly those copies).  `value` stays the resident-input rate the contract asks
    # for; this is what training pays on top (reference call pattern: examples/ScanNet/train_instance.py:345-400).
    h2d = None
    if world == 1 and args.host_steps > 0:
        from occuseg_amd.data import PrefetchingLoader
        host_b = [[b[0].cpu().pin_memory(), b[1].cpu().pin_memory(), None, b[3]] for b in batches]
        SCN.DW_SIDE_STREAM = dw_side
        SCN._PREFETCHED.clear()
        n_host_warm = 2
        loader = PrefetchingLoader(_HostBatches(host_b, 0, n_host_warm + args.host_steps), input_layer, dev)
        loader.record_exposed = True

        def step_on(x):
            bucket.zero()
            o = net(x)
            o.sum().backward()
            bucket.all_reduce_mean()
            opt.step()

        it = iter(loader)
        for _ in range(n_host_warm):
            step_on(next(it)["x"])
        torch.cuda.synchronize()
        th = time.perf_counter()
        sites_h = 0
        for k in range(args.host_steps):
            step_on(next(it)["x"])
            sites_h += n_sites[(n_host_warm + k) % len(batches)]
        torch.cuda.synchronize()
        dth = time.perf_counter() - th
        ex = sorted(e0.elapsed_time(e1) for e0, e1 in loader.exposed[n_host_warm:])
        nbytes = sum(t.numel() * t.element_size() for t in host_b[0][:2])
        h2d = {"ms_per_step_from_host": 1e3 * dth / args.host_steps, "value_from_host": sites_h / dth, "steps": args.host_steps,
               "exposed_ms": ex[len(ex) // 2] if ex else None, "exposed_ms_max": ex[-1] if ex else None,
               "bytes_per_batch": nbytes,
               "how": "inputs pinned on the host; per step the NEXT batch's coordinates + features are copied on a copy stream "
                      "(non-blocking) and its voxel hash + rulebooks built behind the copy on the side stream while this step's "
                      "kernels drain (occuseg_amd.data.PrefetchingLoader); exposed_ms = median time the main stream waited for "
                      "the copy stream"}
        SCN._PREFETCHED.clear()
        SCN.DW_SIDE_STREAM = False
        del loader, it, host_b

    # N > 1: per rank, the time the main stream stood still for the gradient exchange (end of the backward kernels -> last
    # bucket complete; the buckets launched from inside backward overlap it) and the active voxels it processed
    exposed_ms = sorted(e0.elapsed_time(e1) for e0, e1 in bucket.exposed[-(args.steps + 1):-1]) if bucket.exposed else []
    exposed_med = exposed_ms[len(exposed_ms) // 2] if exposed_ms else 0.0
    t = torch.tensor([dt, float(sites_done)], dtype=torch.float64, device=dev)
    per_rank = None
    if world > 1:
        tcpu = t.cpu() if backend == "gloo" else t
        tmax = tcpu.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tcpu.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        mine = torch.tensor([exposed_med, float(sites_done), dt], dtype=torch.float64, device=dev)
        mine = mine.cpu() if backend == "gloo" else mine
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = {"allreduce_exposed_ms": [float(g[0]) for g in gathered],
                    "active_voxels": [float(g[1]) for g in gathered],
                    "seconds": [float(g[2]) for g in gathered]}
        dt, total_sites = float(tmax[0]), float(tsum[1])
    else:
        total_sites = float(sites_done)

    if rank == 0:
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        # per-kernel HIP-event timings over the timed region (events were recorded on the launch stream)
        agg = {}
        launches = []
        sampled = len(range(0, args.steps, args.event_every))  # steps whose matrix launches carry events
        for plist, per in ((prof, 1.0), (prof_extra, float(sampled))):  # the extra step counts as `sampled` steps' worth
            for kernel, flops, nbytes, e0, e1, shape in plist:
                nbytes = nbytes() if callable(nbytes) else nbytes
                ms = e0.elapsed_time(e1)
                a = agg.setdefault(kernel, [0, 0.0, 0.0, 0.0])
                a[0] += per
                a[1] += flops * per
                a[2] += nbytes * per
                a[3] += ms * 1e-3 * per
                launches.append((kernel, shape, ms * per, flops * per, nbytes * per, per))
        peaks = measured_peaks(dev) if world == 1 else None
        hbm_meas = max(peaks["hbm_copy_GBs"], peaks["hbm_read_GBs"]) if peaks else None
        # HBM bytes per launch of every kernel: rocprofv3 PMC counters cannot be read from inside this process; they are
        # collected by running THIS command under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and
        # summarised by tools/pmc_traffic.py, which stamps the file with the hash of the kernel sources it measured.  A file
        # measured on other sources is refused.
        counted, traffic_note = {}, None
        if os.path.exists(TRAFFIC_FILE) and args.m == 64 and args.scenes_per_gpu == 4 and args.sites == 150000:
            try:
                tj = json.load(open(TRAFFIC_FILE))
                if tj.get("source_hash") == source_hash():
                    counted = {k: v["hbm_bytes_per_launch"] for k, v in tj.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
                    traffic_note = ("profiles/r06_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, "
                                    "separate passes; FETCH_SIZE x2 per the gfx950 note), same kernel sources")
                else:
                    traffic_note = "profiles/r06_traffic.json was measured on other kernel sources: refused"
            except (OSError, ValueError, KeyError):
                counted = {}
        kernels = {}
        for kname, (cnt, flops, nbytes, secs) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
            # `hbm_frac`: share of the 8 TB/s the launch's HBM traffic amounts to.  The bytes are the SMALLER of the algorithmic
            # count (SURVEY 8d's formula, which charges e.g. one dOut row per RULE to dWeight although a row that several rules
            # of a workgroup name is fetched once) and the bytes the counters saw -- a fraction above 1 (round 3: dw_stem 1.66)
            # only said that the formula over-counts.  `hbm_bytes`: which of the two it is.
            cb = counted.get(kname, counted.get(kname.replace("_split", "")))
            algo_pl = nbytes / cnt
            use = min(algo_pl, cb) if cb else algo_pl
            k = {"launches_per_step": cnt / sampled, "ms_per_step": 1e3 * secs / sampled, "avg_ms": 1e3 * secs / cnt,
                 "algo_GBs": nbytes / secs / 1e9, "algo_bytes_per_launch": algo_pl,
                 "counted_bytes_per_launch": cb, "counted_over_algorithmic": (cb / algo_pl) if cb else None,
                 # bytes these launches read on top of the formula because a shortcut / a batch norm's backward sums are fused
                 # into them (SCN.PROFILE_FUSED): traffic beyond algorithmic + fused is re-reading, traffic beyond algorithmic is not
                 "fused_extra_bytes_per_launch": SCN.PROFILE_FUSED.get(kname, 0.0) / cnt,
                 "counted_over_algorithmic_plus_fused": (cb / (algo_pl + SCN.PROFILE_FUSED.get(kname, 0.0) / cnt)) if cb else None,
                 "hbm_bytes": "counted" if (cb and cb < algo_pl) else "algorithmic",
                 "hbm_frac": use * cnt / secs / 1e9 / HBM_PEAK_GBS}
            if k["hbm_frac"] > 1.0:  # (only without a counter file for these sources: the formula's over-count, not a rate)
                k["hbm_frac"] = None
                k["hbm_bytes"] = "algorithmic, over-counted (no counter file for these kernel sources)"
            if flops > 0:
                k["tflops"] = flops / secs / 1e12
                k["mfma_frac"] = flops / secs / 1e12 / F32_MFMA_PEAK_TFLOPS
                k["bound"] = "mfma" if flops / max(use * cnt, 1.0) > F32_MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else "hbm"
            if hbm_meas and k["hbm_frac"] is not None:
                k["hbm_frac_of_measured"] = use * cnt / secs / 1e9 / hbm_meas
            kernels[kname] = k
        groups = {"bn": ("bn_fwd", "bn_bwd"), "rules": ("grid_build", "sub_rules", "strided_rules"), "io": ("io",)}
        for g, names in groups.items():  # the HBM-bound parts of the step, summed
            sel = [agg[n] for n in names if n in agg]
            if sel:
                secs, nbytes = sum(a[3] for a in sel), sum(a[2] for a in sel)
                kernels[g] = {"ms_per_step": 1e3 * secs / sampled, "algo_bytes_per_step": nbytes / sampled,
                              "algo_GBs": nbytes / secs / 1e9, "hbm_frac": nbytes / secs / 1e9 / HBM_PEAK_GBS,
                              "members": [n for n in names if n in agg],
                              "measured": "HIP events in one extra step right after the timed region"}
        if "rules" in kernels:
            # inside the step the rule build runs on the side stream UNDER the step's big kernels (that is its point),
            # so the event times above include waiting for CUs.  The same build alone on an idle GPU:
            torch.cuda.synchronize()
            SCN._PREFETCHED.clear()  # really build: do not adopt what the last step prefetched
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            alone = []
            built0 = SCN.OCC_STATS["built"]
            plan_key = (4096, 4096, 4096)
            plan = list(SCN._PLANS_USED.get(plan_key, []))  # the rulebook requests of one forward pass of this network
            chain, SCN.GRID_CHAIN = SCN.GRID_CHAIN, False  # (as prefetch_input builds it in training: a read-back per scale)
            for b in batches + batches:
                SCN._PLANS[plan_key] = list(plan)
                md = SCN.Metadata_3()
                e0.record()
                md.inputLayer(torch.tensor([4096] * 3), b[0], 0, 4, dev)  # grids + every rulebook of the recorded plan
                e1.record()
                torch.cuda.synchronize()
                alone.append(e0.elapsed_time(e1))
            SCN.GRID_CHAIN = chain
            kernels["rules"]["alone_ms_per_batch"] = min(alone)
            kernels["rules"]["alone_ms_all"] = alone
            kernels["rules"]["alone_rulebooks_built"] = len(plan)
            kernels["rules"]["alone_algo_GBs"] = kernels["rules"]["algo_bytes_per_step"] / min(alone) / 1e6
            kernels["rules"]["alone_hbm_frac"] = kernels["rules"]["alone_algo_GBs"] / HBM_PEAK_GBS
            kernels["rules"]["alone_path"] = "occupancy bitmaps" if SCN.OCC_STATS["built"] > built0 else "sort + hash"
            kernels["rules"]["note"] = ("ms_per_step / algo_bytes: the per-scale sort + hash build of the PROFILED extra step (events "
                                        "around every piece keep it off the one-call batch build), on the side stream, overlapped with "
                                        "and slowed by the step's kernels; alone_*: the batch's grids and rulebooks as an InputLayer call "
                                        "builds them from the second batch on (alone_path), on an idle GPU, incl. its two read-backs")
        mm = [k for k in agg if k.startswith(("conv_", "dw_"))]
        dom = max(mm, key=lambda k: agg[k][3]) if mm else None
        traffic = counted.get(dom or "")
        roof = None
        if dom:
            cnt, flops, nbytes, secs = agg[dom]
            achieved = flops / secs / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": traffic_note,
                    "launches_per_step": cnt / sampled, "avg_launch_ms": 1e3 * secs / cnt,
                    "algo_flops_per_launch": flops / cnt, "algo_bytes_per_launch": nbytes / cnt,
                    "algo_GBs": nbytes / secs / 1e9, "hbm_frac_of_8TBs": nbytes / secs / 1e9 / HBM_PEAK_GBS,
                    "durations": ("exclusive: on the %d of %d timed steps whose launches carry events the dWeight kernels run on "
                                  "the main stream; on the others they overlap the kernels that follow them on a side stream "
                                  "(dw_side_stream=%s), where two kernels sharing the chip both take longer"
                                  % (sampled, args.steps, dw_side))}
            if peaks:
                roof["peak_measured"] = peaks["f32_mfma_tflops"]
                roof["frac_of_measured"] = achieved / peaks["f32_mfma_tflops"]
                roof["hbm_peak_measured"] = hbm_meas
            # all gather-mm + dWeight launches together (what round 1 reported as "conv_tile")
            fl, se = sum(agg[k][1] for k in mm), sum(agg[k][3] for k in mm)
            roof["all_matrix_kernels"] = {"tflops": fl / se / 1e12, "frac": fl / se / 1e12 / F32_MFMA_PEAK_TFLOPS,
                                          "ms_per_step": 1e3 * se / sampled}
            # the matrix work of a step over the WHOLE step time (rule build, batch norm, optimizer, gaps included)
            step_s = dt / args.steps
            roof["whole_step"] = {"tflop_per_step": fl / sampled / 1e12, "tflops": fl / sampled / step_s / 1e12,
                                  "frac": fl / sampled / step_s / 1e12 / F32_MFMA_PEAK_TFLOPS}
        res = {
            "metric": "active-voxels/s fwd+bwd, ScanNet m64 UNet, 1/2/4/8 MI355X",
            "value": total_sites / dt, "unit": "active-voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_median": step_ms[len(step_ms) // 2], "ms_per_step_min": step_ms[0],
            # (SURVEY.md section 8d asks for the median of the per-step times; `value` is the driver's contract: all units
            #  over the whole timed region, max over ranks)
            "value_from_median": (total_sites / args.steps) / (step_ms[len(step_ms) // 2] * 1e-3) if world == 1 else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: ScanNet m%d UNet train step (fwd+bwd+Adam), "
                                   "batch=%d synthetic scenes x ~%dk active voxels per GPU, %s, rulebooks rebuilt "
                                   "every step" % (args.m, args.scenes_per_gpu, args.sites // 1000,
                                                   "ONE batch replayed" if args.same_batch else
                                                   "two different batches alternating"),
                       "active_voxels_per_batch": n_sites, "parameters": n_params,
                       "parallelism": "dp%d" % world, "allreduce_bytes": 4 * n_params if world > 1 else 0,
                       "backend": backend if world > 1 else None,
                       "ranks_seen": dist.get_world_size() if world > 1 else 1,
                       "rulebook_prefetch": bool(args.prefetch), "dweight_side_stream": dw_side,
                       "deterministic_dweight": bool(SCN.DETERMINISTIC_DW),
                       "stream_k": ("off" if not SCN.STREAM_K else "forward launches" if SCN.STREAM_K_FWD_ONLY else "all launches"),
                       "stream_k_gave_up": bool(SCN.stream_k_gave_up()),
                       "weight_repack": "every (weight, orientation) pair re-laid out once per step in one launch "
                                        "(SCN weight epochs): %d multi-tensor launches, %d single ones in the timed region"
                                        % (wpack_after["batch"] - wpack_before["batch"], wpack_after["single"] - wpack_before["single"]),
                       "python_gc": "gc.collect() + gc.freeze() behind the warm-up, before the timed region (a generational collection "
                                    "takes 35-40 ms in this process wherever the allocation counts put it)",
                       "rule_build": ("occupancy bitmaps (scn_batch_build_occ)" if SCN.OCC_STATS["built"] else "sort + hash"),
                       "source_hash": source_hash()},
            "roofline": roof, "kernels": kernels, "peaks_measured": peaks,
        }
        if h2d is not None:
            h2d["delta_ms_vs_resident"] = h2d["ms_per_step_from_host"] - res["ms_per_step"]
            res["h2d"] = h2d
        if per_rank is not None:
            # what the first multi-GPU run has to explain about itself: how long each rank's main stream waited for the
            # gradient exchange per step (median), and how evenly the voxels fell (scenes differ by +-5 % in sites)
            res["allreduce_exposed_ms"] = max(per_rank["allreduce_exposed_ms"])
            res["per_rank"] = per_rank
        # active sites and measured nu (pairs per site of the 3x3x3 rulebook) per UNet level, of the
        # batch of the extra step; the ten matrix launches (kernel, shape) that take the most time
        if seen_md:
            md = seen_md[-1]
            lv = []
            for key, rb in md.submanifoldRuleBooks.items():
                n = md.nActive.get(key[0] if isinstance(key[0], tuple) else key[:3])
                if n:
                    lv.append({"spatial_size": int((key[0] if isinstance(key[0], tuple) else key)[0]), "active_sites": int(n),
                               "nu": float(rb.total_pairs) / float(n)})
            res["levels"] = sorted(lv, key=lambda d: -d["active_sites"])
        per_shape = {}
        for kernel, shape, ms, flops, nbytes, per in launches:
            if flops > 0:
                d = per_shape.setdefault((kernel, tuple(shape) if shape else ()), [0.0, 0.0, 0.0])
                d[0] += per
                d[1] += ms
                d[2] += flops
        res["layers"] = [{"kernel": k[0], "K_rows_Ci_Co_transposed_incol": list(k[1]), "launches_per_step": v[0] / sampled,
                          "ms_per_launch": v[1] / v[0], "tflops": v[2] / v[1] / 1e9}
                         for k, v in sorted(per_shape.items(), key=lambda kv: -kv[1][1])[:10]]
        if args.dump_launches:
            per_shape = {}
            for kernel, shape, ms, flops, nbytes, per in launches:
                d = per_shape.setdefault("%s %s" % (kernel, shape), [0, 0.0, 0.0, 0.0])
                d[0] += per
                d[1] += ms
                d[2] += flops
                d[3] += nbytes
            rows = [{"launch": k, "per_step": v[0] / sampled, "ms_per_step": v[1] / sampled, "avg_ms": v[1] / v[0],
                     "tflops": v[2] / v[1] / 1e9 if v[2] else None, "algo_GBs": v[3] / v[1] / 1e6}
                    for k, v in sorted(per_shape.items(), key=lambda kv: -kv[1][1])]
            # ... and every matrix launch of the profiled steps in issue order (which launch of a shape is the slow one)
            seq = [{"kernel": k, "shape": list(sh) if sh else None, "ms": ms / per, "tflops": fl / ms / 1e9 if fl else None,
                    "bytes": nb / per}
                   for k, sh, ms, fl, nb, per in launches if fl]
            json.dump({"per_shape": rows, "in_order": seq}, open(args.dump_launches, "w"), indent=0)
        if world == 1 and not args.no_configs and args.m == 64:
            # BASELINE.json configs[1] and configs[4] in the driver's own line (the workload of `value` stays configs[2])
            del batches[:]
            torch.cuda.empty_cache()
            t_cfg = time.perf_counter()
            res["configs"] = other_configs(scn, dev)
            res["configs"]["seconds"] = time.perf_counter() - t_cfg
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.m, args.scenes_per_gpu, args.sites, 1000, args.cpu_iters, dev_pass=dev_pass)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            # the comparison the CPU leg makes on its way (the reference's compiled loops against the HIP pass taken before the
            # warm-up), lifted to the top level of the line
            res["parity_fullsize"] = res["cpu_baseline"].pop("parity_fullsize", None)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
