"""Child process of tests/test_gpu_dp.py::test_rccl_world1_stream_ordering: a process group of ONE rank with backend "nccl"
(= RCCL on ROCm) on the one GPU a test box has.  No byte crosses xGMI, but RCCL is loaded and its all-reduce kernels run on
the process group's own stream, ordered against the three streams of a train step (main, dWeight side stream, rule-build
prefetch stream) exactly as in an 8-GPU run -- which the gloo tests never did.  The real sparse UNet (m32 x 4 levels) runs
three steps with FlatGradBucket.enable_overlap(): every bucket's exchange is launched from inside backward on the exchange
stream, and the gradients must be BITWISE those of the same step without a process group.  Writes a JSON verdict to argv[1]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_path = sys.argv[1]
    res = {"ok": False}
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        import occuseg_amd
        from occuseg_amd import SCN, dp, scenes
        scn = occuseg_amd.install()
        assert SCN.DW_SIDE_STREAM and SCN.DETERMINISTIC_DW
        torch.manual_seed(0)
        m = 32
        net = scn.Sequential().add(scn.InputLayer(3, 4096, mode=4)).add(
            scn.SubmanifoldConvolution(3, 3, m, 3, False)).add(
            scn.UNet(3, 1, [m, 2 * m, 3 * m, 4 * m], residual_blocks=True)).add(
            scn.BatchNormReLU(m)).add(scn.OutputLayer(3)).to(dev)
        input_layer = next(mod for mod in net.modules() if isinstance(mod, scn.InputLayer))
        batches = []
        for base in (1000, 1100):
            coords, feats = scenes.batch([base, base + 1], target_sites=40000)
            batches.append([torch.from_numpy(coords).to(dev), torch.from_numpy(feats).to(dev), None, 2])
        bucket = dp.FlatGradBucket(net.parameters(), n_buckets=4)
        res["n_buckets"] = len(bucket.buckets)

        # reference gradients: the same steps, no process group (nothing is exchanged, everything else as in training)
        want = []
        for i in range(3):
            bucket.zero()
            net(batches[i % 2]).sum().backward()
            bucket.all_reduce_mean()
            input_layer.prefetch(batches[(i + 1) % 2], wait_for_current_stream=False)
            torch.cuda.synchronize()
            want.append(bucket.flat.clone())
        SCN._PREFETCHED.clear()

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        res["backend"] = dist.get_backend()
        bucket.always_exchange = True
        bucket.enable_overlap()
        bucket.record_exposed = True
        got, logs, joins = [], [], []
        for i in range(3):
            bucket.zero()
            net(batches[i % 2]).sum().backward()
            logs.append(list(bucket.launch_log))
            joins.append(list(bucket.join_log))
            bucket.all_reduce_mean()
            input_layer.prefetch(batches[(i + 1) % 2], wait_for_current_stream=False)  # the next batch's rule build, side stream
            got.append(bucket.flat.clone())  # (main stream: ordered behind the exchange by all_reduce_mean's waits)
        torch.cuda.synchronize()
        res["launched_in_backward"] = [len(lg) for lg in logs]
        res["each_once"] = all(sorted(b for b, _ in lg) == list(range(len(bucket.buckets))) for lg in logs)
        res["complete_at_launch"] = all(set(range(*bucket.buckets[b][2:])) <= rep for lg in logs for b, rep in lg)
        res["exchange_stream_launches"] = [len(j) for j in joins]
        res["bitwise_equal"] = [bool(torch.equal(g, w)) for g, w in zip(got, want)]
        res["max_abs_diff"] = [float((g - w).abs().max()) for g, w in zip(got, want)]
        res["grad_nonzero"] = bool(want[0].abs().max() > 0)
        res["exposed_ms"] = [e0.elapsed_time(e1) for e0, e1 in bucket.exposed]
        res["stream_k_gave_up"] = bool(SCN.stream_k_gave_up())
        res["ok"] = (all(res["bitwise_equal"]) and res["each_once"] and res["complete_at_launch"] and res["grad_nonzero"]
                     and all(n == len(bucket.buckets) for n in res["launched_in_backward"])
                     and all(n == len(bucket.buckets) for n in res["exchange_stream_launches"])
                     and not res["stream_k_gave_up"])
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the parent test prints this
        import traceback
        res["error"] = "%s\n%s" % (e, traceback.format_exc())
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
