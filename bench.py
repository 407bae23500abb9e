#!/usr/bin/env python3
"""bench.py -- active-voxels/s of one fwd+bwd train step of the ScanNet m64 UNet backbone.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: InputLayer (voxel hash + every rulebook is
rebuilt each step, as in real training) -> stem SC(3->m) -> scn.UNet(3,1,[m..6m],residual) ->
BatchNormReLU -> OutputLayer -> sum loss -> backward -> (N>1: RCCL all-reduce of the flat
gradient) -> Adam step.  Workload = BASELINE.json configs[2]: 4 synthetic ScanNet-like scenes of
~150k active voxels per GPU (weak scaling: configs[3] is 8 x this).  Inputs are resident in HBM
before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def build_net(scn, m):
    return scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
        scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
        scn.UNet(3, 1, [m, 2 * m, 3 * m, 4 * m, 5 * m, 6 * m], residual_blocks=True)).add(
        scn.BatchNormReLU(m)).add(scn.OutputLayer(3))


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


def cpu_baseline(m, target_sites, seed, budget_s=20.0):
    """The oracle (numpy rulebooks + torch CPU index_select/mm/index_add_, float32, all host
    cores) on a BOUNDED sample of the same workload: one scene of the same generator, one
    fwd+bwd of the same network, sized (after a 10k-voxel probe) to ~budget_s seconds of CPU
    work and capped at the full scene.  Rulebook construction time is excluded (the numpy rule
    generator is not the reference's algorithm)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--scenes-per-gpu", type=int, default=4)
    ap.add_argument("--sites", type=int, default=150000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=4, help="time the kernels with HIP events on every n-th step")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="build each step's rulebooks inside the step instead of one step ahead on a side stream")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import occuseg_amd
    from occuseg_amd import SCN, dp, scenes
    scn = occuseg_amd.install()

    torch.manual_seed(0)  # identical replicas
    net = build_net(scn, args.m).to(dev)
    n_params = sum(p.numel() for p in net.parameters())
    bucket = dp.FlatGradBucket(net.parameters())
    if world > 1:
        bucket.enable_overlap()  # bucketed all-reduce launched from backward hooks
    # torch's fused Adam kernel (same arithmetic as the default foreach implementation: 53.5 -> 51.9 ms per step)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=bool(int(os.environ.get("SCN_BENCH_FUSED_ADAM", "1"))))

    all_seeds = [1000 + i for i in range(args.scenes_per_gpu * world)]
    seeds = dp.shard(all_seeds, rank, world)
    coords, feats = scenes.batch(seeds, args.sites)
    coords_d = torch.from_numpy(coords).to(dev)
    feats_d = torch.from_numpy(feats).to(dev)
    batch_in = [coords_d, feats_d, None, len(seeds)]

    with torch.no_grad():
        md = SCN.Metadata_3()
        md.inputLayer(torch.tensor([4096] * 3), coords_d, 0, 4, dev)
        n_sites = md.nActive[md.inputSpatialSize]
        del md
    input_layer = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))

    def step():
        bucket.zero()
        out = net(batch_in)
        out.sum().backward()
        bucket.all_reduce_mean()
        opt.step()
        if args.prefetch:
            # software pipelining across steps: the NEXT batch's voxel hash + rulebooks are built on a side stream
            # while this step's kernels drain (every step still builds its own)
            input_layer.prefetch(batch_in, wait_for_current_stream=False)
        return out

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every gather-mm / dWeight launch, inside the timed region, on every EVENT_EVERY-th step: a
    # timed event drains the queue between two kernels, and 226 of them per step cost 3-4 ms of a 57 ms step
    # (measured back to back: 58.8 / 57.5 ms with events on every step, 54.7 / 54.3 ms with none).
    prof = []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        SCN.PROFILE = prof if (i % args.event_every == 0 and rank == 0) else None
        step()
    SCN.PROFILE = None
    barrier()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt, float(n_sites)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_sites = float(tmax[0]), float(tsum[1])
    else:
        total_sites = float(n_sites)

    if rank == 0:
        # per-kernel HIP-event timings over the timed region (events were recorded on the launch stream)
        agg = {}
        for kernel, flops, nbytes, e0, e1, _shape in prof:
            a = agg.setdefault(kernel, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += nbytes
            a[3] += e0.elapsed_time(e1) * 1e-3
        kernels = {}
        sampled = len(range(0, args.steps, args.event_every))  # steps whose launches carry events
        for kname, (cnt, flops, nbytes, secs) in agg.items():
            kernels[kname] = {"launches": cnt, "avg_ms": 1e3 * secs / cnt, "tflops": flops / secs / 1e12,
                              "algo_GBs": nbytes / secs / 1e9, "share_of_step": secs / (dt * sampled / args.steps),
                              "steps_with_events": sampled}
        dom = max(agg, key=lambda k: agg[k][3]) if agg else None
        # HBM bytes per launch of the dominant kernel: rocprofv3 PMC counters cannot be read from inside
        # this process; they are collected by running THIS command under `rocprofv3 --pmc FETCH_SIZE` and
        # `--pmc WRITE_SIZE` (separate passes) and summarised by tools/pmc_traffic.py into profiles/.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.m == 64 and args.scenes_per_gpu == 4 and args.sites == 150000:
            try:
                traffic = json.load(open(tpath)).get(dom or "", {}).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        roof = None
        if dom:
            cnt, flops, nbytes, secs = agg[dom]
            achieved = (flops / cnt) / (secs / cnt) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this "
                                      "command; FETCH_SIZE x2 per the gfx950 note)" if traffic else None,
                    "launches": cnt, "avg_launch_ms": 1e3 * secs / cnt,
                    "algo_flops_per_launch": flops / cnt, "algo_bytes_per_launch": nbytes / cnt,
                    "algo_GBs": nbytes / secs / 1e9, "hbm_frac_of_8TBs": nbytes / secs / 1e9 / HBM_PEAK_GBS}
        res = {
            "metric": "active-voxels/s fwd+bwd, ScanNet m64 UNet, 1/2/4/8 MI355X",
            "value": total_sites * args.steps / dt, "unit": "active-voxels/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: ScanNet m%d UNet train step (fwd+bwd+Adam), "
                                   "batch=%d synthetic scenes x ~%dk active voxels per GPU, rulebooks rebuilt "
                                   "every step" % (args.m, args.scenes_per_gpu, args.sites // 1000),
                       "active_voxels_per_step": total_sites, "parameters": n_params,
                       "parallelism": "dp%d" % world, "allreduce_bytes": 4 * n_params if world > 1 else 0,
                       "rulebook_prefetch": bool(args.prefetch)},
            "roofline": roof, "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.m, args.sites, 1000)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
