"""not gpu: the N>1 path on CPU -- two processes, gloo backend.  Checks the flat gradient bucket
(one all-reduce of one contiguous buffer, averaged) and the scene sharding used by bench.py."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from occuseg_amd import dp
    torch.manual_seed(0)  # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    bucket = dp.FlatGradBucket(net.parameters(), n_buckets=2)
    assert len(bucket.buckets) == 2 and bucket.buckets[0][0] == 0 and bucket.buckets[-1][1] == bucket.flat.numel()
    if ret.get("overlap"):
        bucket.enable_overlap()
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    seeds = dp.shard(list(range(100, 100 + 4 * world)), rank, world)
    assert seeds == list(range(100 + 4 * rank, 104 + 4 * rank))
    g = torch.Generator().manual_seed(seeds[0])
    x = torch.randn(7, 6, generator=g)
    bucket.zero()
    net(x).sum().backward()
    if ret.get("overlap"):
        assert len(bucket._launched) == 2  # both buckets were sent from backward hooks
        # ... and each only after EVERY one of its parameters had reported its gradient
        for b, reported in bucket.launch_log:
            first, last = bucket.buckets[b][2:]
            assert set(range(first, last)) <= reported, (b, sorted(reported))
        local = None
    else:
        local = bucket.flat.clone()
    if local is None:  # recompute the un-reduced gradient for the check
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        ref.load_state_dict(net.state_dict())
        ref(x).sum().backward()
        local = torch.cat([p.grad.flatten() for p in ref.parameters()])
    bucket.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = torch.allclose(bucket.flat, want, atol=1e-7) and all(
        torch.equal(p.grad.flatten(), bucket.flat[o:o + p.numel()])
        for p, o in zip(net.parameters(), [0, 30, 35, 50]))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("overlap", [False, True])
def test_flat_gradient_allreduce_world2(overlap):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        ret["overlap"] = overlap
        mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert ret[0] is True and ret[1] is True


def test_bucket_waits_for_every_parameter_whatever_the_order():
    """The advisor's case: the FIRST parameter of a bucket reports before the others (a convolution weight whose
    bias gradient arrives later through autograd).  The bucket must not launch until all of them have reported, and a
    report after the launch is an error."""
    from occuseg_amd import dp
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    b = dp.FlatGradBucket(net.parameters(), n_buckets=2)
    b.enable_overlap()
    params = list(net.parameters())
    first, last = b.buckets[0][2:]
    assert last - first >= 2
    b.zero()
    b.grad_written(params[first])          # direct accumulation reports the first parameter of bucket 0 ...
    assert 0 not in b._launched            # ... the bucket is NOT complete yet
    for i in range(first + 1, last):
        b.grad_written(params[i])
    assert 0 in b._launched and b.launch_log[0][1] >= set(range(first, last))
    with pytest.raises(RuntimeError, match="written twice in one step"):
        b.grad_written(params[first])
    b.zero()
    b._written(first, "hook")              # autograd's hook following the direct report of the same write: fine
    b.grad_written(params[first])
    b.close()
    assert not hasattr(params[0], "_scn_grad_owner")


def test_bucket_is_a_noop_without_process_group():
    from occuseg_amd import dp
    net = torch.nn.Linear(3, 2)
    b = dp.FlatGradBucket(net.parameters())
    net(torch.ones(1, 3)).sum().backward()
    before = b.flat.clone()
    b.all_reduce_mean()
    assert torch.equal(before, b.flat) and b.flat.abs().sum() > 0
