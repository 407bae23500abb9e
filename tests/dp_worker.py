"""Child process of tests/test_gpu_dp.py: one rank of a world-size-2 process group (gloo) in which BOTH ranks use
cuda:0 -- the one GPU a test box has -- and run the real sparse UNet (mirror's scn.UNet, m16 x 3 levels) on their own
two scenes with FlatGradBucket.enable_overlap().  Writes a JSON verdict to argv[1].<rank>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    res = {"rank": rank, "ok": False}
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        import occuseg_amd
        from occuseg_amd import dp, scenes
        scn = occuseg_amd.install()
        torch.manual_seed(0)  # identical replicas
        m = 16
        net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
            scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
            scn.UNet(3, 1, [m, 2 * m, 3 * m], residual_blocks=True)).add(
            scn.BatchNormReLU(m)).add(scn.OutputLayer(3)).to(dev)
        seeds = dp.shard([1000, 1001, 1002, 1003], rank, world)
        coords, feats = scenes.batch(seeds, target_sites=5000)
        batch = [torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, len(seeds)]
        bucket = dp.FlatGradBucket(net.parameters(), n_buckets=4)
        res["n_buckets"] = len(bucket.buckets)
        running = [b.clone() for b in net.buffers()]

        # pass 1, no overlap: the un-reduced local gradient, then the mean over ranks as the reference value
        bucket.zero()
        net(batch).sum().backward()
        local = bucket.flat.clone()
        gathered = [torch.zeros_like(local).cpu() for _ in range(world)]
        dist.all_gather(gathered, local.cpu())
        want = sum(gathered) / world
        res["ranks_differ"] = bool((gathered[0] - gathered[1]).abs().max() > 0)
        for b, r in zip(net.buffers(), running):  # same batch-norm state for pass 2
            b.copy_(r)

        # pass 2, overlapped: every bucket launched from backward, exactly once, only after ALL its parameters reported
        bucket.enable_overlap()
        bucket.zero()
        net(batch).sum().backward()
        log = list(bucket.launch_log)
        res["launched_in_backward"] = len(log)
        res["each_once"] = sorted(b for b, _ in log) == list(range(len(bucket.buckets)))
        res["complete_at_launch"] = all(set(range(*bucket.buckets[b][2:])) <= rep for b, rep in log)
        # backward order is not parameter order: the first parameter of some bucket is NOT the last to report
        bucket.all_reduce_mean()
        got = bucket.flat.cpu()
        scale = float(want.abs().max())
        res["max_err_rel"] = float((got - want).abs().max()) / scale
        res["views_intact"] = all(bucket.owns(p) for p in net.parameters())
        res["ok"] = (res["each_once"] and res["complete_at_launch"] and res["views_intact"]
                     and res["max_err_rel"] < 1e-5 and res["launched_in_backward"] == len(bucket.buckets))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the parent test prints this
        import traceback
        res["error"] = "%s\n%s" % (e, traceback.format_exc())
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
