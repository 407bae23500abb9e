"""-m gpu: the loss-side hot spots SURVEY.md section 8 f4 names -- the per-instance loop of calculate_cost
(examples/ScanNet/train_instance.py:204-254) and the region-based mean-shift clustering of the evaluation
(examples/ScanNet/evaluate_instance.py:646-806) -- against loop-for-loop restatements (oracle/losses.py)."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _batch(rng, scenes=3, pts=4000, max_inst=17):
    bidx, inst, sem, sizes = [], [], [], []
    for c in range(scenes):
        n = pts + 100 * c
        k = int(rng.integers(1, max_inst))
        ids = rng.integers(0, k, n)
        ids[:k] = np.arange(k)  # every id present (the collate function renumbers 0..k-1)
        cls_of = rng.integers(0, 6, k)  # classes 0/1 = stuff (wall, floor): skipped by the loss
        lab = cls_of[ids].copy()
        noise = rng.random(n) < 0.1
        lab[noise] = rng.integers(0, 6, int(noise.sum()))  # the FIRST point decides, not the majority
        bidx.append(np.full(n, c)), inst.append(ids), sem.append(lab), sizes.append(k)
    return np.concatenate(bidx), np.concatenate(inst), np.concatenate(sem), sizes


def test_instance_regression_losses(gpu):
    from occuseg_amd import losses
    from oracle import losses as O
    rng = np.random.default_rng(0)
    bidx, inst, sem, sizes = _batch(rng)
    P = len(bidx)
    disp, disp_gt = rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, 3)).astype(np.float32)
    occ, occ_gt = rng.normal(size=(P, 1)).astype(np.float32) + 3, rng.normal(size=(P, 1)).astype(np.float32) + 3
    want_d, want_o = O.instance_regression_losses(disp, disp_gt, occ, occ_gt, sem, bidx, inst)
    d = torch.from_numpy(disp).to(gpu).requires_grad_(True)
    o = torch.from_numpy(occ).to(gpu).requires_grad_(True)
    args = (torch.from_numpy(disp_gt).to(gpu), o, torch.from_numpy(occ_gt).to(gpu), torch.from_numpy(sem).to(gpu),
            torch.from_numpy(bidx).to(gpu), torch.from_numpy(inst).to(gpu), sizes)
    got_d, got_o = losses.instance_regression_losses(d, *args)
    assert abs(got_d.item() - want_d) <= 1e-4 * abs(want_d) and abs(got_o.item() - want_o) <= 1e-4 * abs(want_o)
    # gradients against float64 autograd over the same formulas written with index_add
    (got_d + 2.0 * got_o).backward()
    d64 = torch.from_numpy(disp).double().requires_grad_(True)
    o64 = torch.from_numpy(occ).double().requires_grad_(True)
    total = 0.0
    for c in range(len(sizes)):
        index = torch.from_numpy(bidx == c)
        ii = torch.from_numpy(inst[bidx == c])
        k = sizes[c]
        cnt = torch.zeros(k, dtype=torch.float64).index_add_(0, ii, torch.ones(len(ii), dtype=torch.float64))
        dn = torch.norm(d64[index] - torch.from_numpy(disp_gt[bidx == c]).double(), dim=1)
        on = torch.norm(o64[index] - torch.from_numpy(occ_gt[bidx == c]).double(), dim=1)
        dm = torch.zeros(k, dtype=torch.float64).index_add(0, ii, dn) / cnt
        om = torch.zeros(k, dtype=torch.float64).index_add(0, ii, on) / cnt
        mean = torch.zeros(k, dtype=torch.float64).index_add(0, ii, o64[index][:, 0]) / cnt
        var = torch.zeros(k, dtype=torch.float64).index_add(0, ii, (o64[index][:, 0] - mean[ii]) ** 2) / (cnt - 1).clamp(min=1)
        std = torch.sqrt(var)
        first_cls = np.array([sem[bidx == c][inst[bidx == c] == m][0] for m in range(k)])
        valid = torch.from_numpy(first_cls > 1)
        if valid.any():
            total = total + (dm[valid].sum() + 2.0 * (om[valid] + std[valid]).sum()) / int(valid.sum()) / len(sizes)
    total.backward()
    assert_close(d.grad.cpu().numpy(), d64.grad.numpy(), "d loss / d displacements")
    assert_close(o.grad.cpu().numpy(), o64.grad.numpy(), "d loss / d occupancy")
    # a scene without any foreground instance contributes nothing but still counts in the batch size
    sem2 = sem.copy()
    sem2[bidx == 1] = 0
    want_d2, want_o2 = O.instance_regression_losses(disp, disp_gt, occ, occ_gt, sem2, bidx, inst)
    got_d2, got_o2 = losses.instance_regression_losses(d.detach(), args[0], o.detach(), args[2],
                                                       torch.from_numpy(sem2).to(gpu), *args[4:])
    assert abs(got_d2.item() - want_d2) <= 1e-4 * abs(want_d2) and abs(got_o2.item() - want_o2) <= 1e-4 * abs(want_o2)


def _scene_for_clustering(rng, n_inst=7, regions_per=9, pts_per_region=60, sem_len=8):
    """well-separated instances (embedding + position), each made of several regions (supervoxels)"""
    emb, xyz, disp, sem, reg, bw, off = [], [], [], [], [], [], []
    rid = 0
    for i in range(n_inst + 2):  # the last two "instances" are stuff (semantic 0 / 1)
        centre_e = rng.normal(size=sem_len) * 4
        centre_p = rng.uniform(0, 8, 3)
        label = (i % 5) + 2 if i < n_inst else i - n_inst
        for _ in range(regions_per):
            n = pts_per_region + int(rng.integers(0, 20))
            p = centre_p + rng.normal(size=(n, 3)) * 0.3
            emb.append(centre_e + rng.normal(size=(n, sem_len)) * 0.05)
            xyz.append(p)
            disp.append(p - centre_p + rng.normal(size=(n, 3)) * 0.02)  # xyz - displacement ~ the instance centre
            sem.append(np.full(n, label))
            reg.append(np.full(n, 1000 + 7 * rid))  # region ids need not be contiguous
            bw.append(np.tile([0.6, 0.9], (n, 1)) + rng.normal(size=(n, 2)) * 0.01)
            off.append(np.exp(-np.linalg.norm(p - centre_p, axis=1) ** 2)[:, None])
            rid += 1
    cat = np.concatenate
    perm = rng.permutation(sum(len(e) for e in emb))
    return [cat(a)[perm].astype(np.float32) if a is not sem and a is not reg else cat(a)[perm]
            for a in (off, emb, sem, disp, xyz, bw, reg)]


def test_region_meanshift_clustering(gpu):
    from occuseg_amd import cluster
    from oracle import losses as O
    rng = np.random.default_rng(3)
    off, emb, sem, disp, xyz, bw, reg = _scene_for_clustering(rng)
    want = O.region_meanshift(off, emb, sem, disp, xyz, bw, reg, sem_len=emb.shape[1])
    got = cluster.region_meanshift(off, emb, sem, disp, xyz, bw, reg, sem_len=emb.shape[1], device=gpu)
    assert got.dtype == np.int32 and got.shape == want.shape
    assert np.array_equal(got, want)
    # the instances found are the generated ones: one label per instance, stuff = -1
    assert set(np.unique(got[sem < 2])) == {-1} and len(np.unique(got[sem >= 2])) == 7
