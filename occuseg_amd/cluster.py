"""Region-based cross-modal mean-shift clustering on the device (the evaluation-side hot spot).

Replaces ``region_based_cross_modal_meanshift_clustering`` (examples/ScanNet/evaluate_instance.py:646-806; helpers
:113-160), which the reference's evaluation calls once per scene (:1407).  The reference keeps the per-region tensors on
the GPU but drives the algorithm from the host: one Python loop over the regions with a boolean mask over all points for
the semantic mode (:683-684), and inside the instance loop a host synchronisation per mean-shift iteration (the
``if norm < 1e-5: break``, :719) plus several per instance (``stats.mode`` on a tensor copied to the host, :722;
``.sum()`` tests).  Here:
  * the per-region pre-work is segment reductions (occuseg_amd.segment: HIP kernels) and one [regions x classes] count;
  * an instance costs ONE host synchronisation (the ``background_mask.sum() > 0`` test of the while loop): the
    data-dependent ``break`` is a device-side "done" flag that freezes the state, the mode is an index_add + argmax.
Same arithmetic (float32 on the GPU, as the reference), same order of instances, same labels.
"""
import math

import numpy as np
import torch

from . import segment


def gaussian_weighted_region_mean(emb, region_mask, n):
    """evaluate_instance.py:119-132"""
    centers = segment.scatter_mean(emb, region_mask, dim=0, dim_size=n)
    std = torch.norm(segment.scatter_std(emb, region_mask, dim=0, dim_size=n) + 1e-8, dim=1)[region_mask]
    w = torch.exp(-(torch.norm(emb - centers[region_mask], dim=1) / std) ** 2)
    wsum = segment.scatter_add(w, region_mask, dim=0, dim_size=n)
    return segment.scatter_add(emb * w.view(-1, 1), region_mask, dim=0, dim_size=n) / wsum.view(-1, 1)


def _cross_modal_gaussian(dist, sem_len, b1, b2):
    d1 = torch.norm(dist[:, :sem_len], dim=1) * b1
    d2 = torch.norm(dist[:, sem_len:sem_len + 3], dim=1) * b2
    return torch.exp(-d1 * d1 - d2 * d2)


def _gaussian(d, bw):
    return torch.exp(-0.5 * (d / bw) ** 2) / (bw * math.sqrt(2 * math.pi))


def region_meanshift(existing_offsets, pred_embedding, pred_semantic, pred_displacements, xyz, pred_bw, regions,
                     bw_relax_factor=1.0, sem_len=None, expected_instance_num=300, max_iter=10, device="cuda"):
    """arguments as the reference's function (numpy arrays or tensors, one scene); returns int32 numpy [P]: the instance
    id of every point, -1 = background / stuff classes (semantic < 2)."""
    dev = torch.device(device)

    def t(a, dtype=torch.float32):
        return torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dev, dtype)
    emb = t(pred_embedding)
    pose = t(xyz) - t(pred_displacements)
    sem = t(pred_semantic, torch.int64)
    P = emb.size(0)
    sem_len = emb.size(1) if sem_len is None else int(sem_len)
    _uniq, region_mask = torch.unique(t(regions, torch.int64), sorted=True, return_inverse=True)
    n = int(_uniq.numel())
    E = torch.cat([gaussian_weighted_region_mean(emb, region_mask, n),
                   gaussian_weighted_region_mean(pose, region_mask, n)], 1)
    BW = gaussian_weighted_region_mean(t(pred_bw), region_mask, n)
    n_cls = int(sem.max().item()) + 1 if P else 1
    votes = torch.zeros(n * n_cls, dtype=torch.float32, device=dev).index_add_(
        0, region_mask * n_cls + sem, torch.ones(P, dtype=torch.float32, device=dev)).view(n, n_cls)
    center_sem = votes.argmax(1)  # stats.mode: the smallest of the most frequent labels = the first maximum
    npts = votes.sum(1)
    off = segment.scatter_max(t(existing_offsets).view(P, -1), region_mask, dim=0, dim_size=n)[0][:, 0].clone()
    background = center_sem >= 2
    pred = torch.where(background, 0, -1).to(torch.int32)
    off = torch.where(background, off, torch.zeros_like(off))
    count = 0
    eye_cls = torch.arange(n_cls, device=dev)
    while count < expected_instance_num and bool(background.any()):  # the one host synchronisation per instance
        key = torch.argmax(off)
        pre = E[key]
        b = BW[key] * bw_relax_factor
        dist = E - pre
        flag = (_cross_modal_gaussian(dist, sem_len, b[0], b[1]) > 0.5) & background
        w = _gaussian(torch.norm(dist, dim=1), b.mean()) * npts * flag
        pre = (E * w.view(-1, 1)).sum(0) / w.sum()
        done = torch.zeros((), dtype=torch.bool, device=dev)
        for _ in range(max_iter):
            dist = E - pre
            fl = flag.to(torch.float32)
            b = (BW * fl.view(-1, 1)).sum(0) / fl.sum() * bw_relax_factor
            new_flag = (_cross_modal_gaussian(dist, sem_len, b[0], b[1]) > 0.5) & background
            w = _gaussian(torch.norm(dist, dim=1), b.mean()) * npts * new_flag
            centroid = (E * w.view(-1, 1)).sum(0) / w.sum()
            shift = torch.norm(pre - centroid)
            upd = ~done  # (the reference updates its state in the converging iteration, then breaks)
            pre = torch.where(upd, centroid, pre)
            flag = torch.where(upd, new_flag, flag)
            done = done | (upd & (shift < 1e-5))
        cls_votes = torch.zeros(n_cls, dtype=torch.float32, device=dev).index_add_(0, center_sem, flag.to(torch.float32))
        label = eye_cls[cls_votes.argmax()]
        flag = flag & (center_sem == label)
        background = background & ~flag
        off = torch.where(flag, torch.zeros_like(off), off)
        count += 1
        pred = torch.where(flag, torch.full_like(pred, count), pred)
    return pred[region_mask].cpu().numpy()
