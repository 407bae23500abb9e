"""Oracle: the per-instance loops of the reference's loss and its mean-shift clustering, restated with numpy / plain
Python loops.  TEST INFRASTRUCTURE.  parity unpinned by reference outputs (the reference holds no tests or fixtures for
these functions, and they import torch_scatter / open3d / scipy.stats, absent or unusable here); each function follows
the cited lines statement by statement."""
import math

import numpy as np

from . import scatter as S_


def instance_regression_losses(displacements, displacements_gt, occupancy, occupancy_gt, semantics, batch_index,
                               instance_mask):
    """examples/ScanNet/train_instance.py:204-254: the loop over scenes (`for count, idx in enumerate(tbl)`) and over
    instances (`for mid in range(mask_size)`), DisplacementLoss and OccupancyLoss as returned (after `/= batchSize`)."""
    disp_loss = occ_loss = 0.0
    batch_size = 0
    for count in range(int(batch_index.max()) + 1):
        index = batch_index == count
        inst = instance_mask[index].astype(np.int64)
        sem = semantics[index]
        d_cluster = S_.scatter_mean(np.linalg.norm(displacements[index] - displacements_gt[index], axis=1), inst)
        o_cluster = S_.scatter_mean(np.linalg.norm(occupancy[index] - occupancy_gt[index], axis=1), inst)
        o_std = S_.scatter_std(occupancy[index], inst)
        d_err = o_err = 0.0
        cluster_size = 0
        for mid in range(int(inst.max()) + 1):
            cls = sem[inst == mid][0]
            if cls > 1:
                d_err += d_cluster[mid]
                o_err += o_cluster[mid] + float(np.sum(o_std[mid]))
                cluster_size += 1
        if cluster_size > 0:
            occ_loss += o_err / cluster_size
            disp_loss += d_err / cluster_size
        batch_size += 1
    return disp_loss / batch_size, occ_loss / batch_size


# ---- examples/ScanNet/evaluate_instance.py:113-160 ------------------------------------------------------------------
def _weighted_mean(emb, w):
    return (emb * w[:, None]).sum(0) / w.sum()


def _gaussian(d, bw):
    return np.exp(-0.5 * (d / bw) ** 2) / (bw * math.sqrt(2 * math.pi))


def gaussian_weighted_region_mean(emb, region_mask, n):
    centers = S_.scatter_mean(emb, region_mask, n)
    std = np.linalg.norm(S_.scatter_std(emb, region_mask, n) + 1e-8, axis=1)[region_mask]
    w = np.exp(-(np.linalg.norm(emb - centers[region_mask], axis=1) / std) ** 2)
    return S_.scatter_add(emb * w[:, None], region_mask, n) / S_.scatter_add(w, region_mask, n)[:, None]


def _cross_modal_gaussian(d1, d2, b1, b2):
    d1 = np.linalg.norm(d1, axis=1) * b1
    d2 = np.linalg.norm(d2, axis=1) * b2
    return np.exp(-d1 * d1 - d2 * d2)


def _mode(values):
    """scipy.stats.mode: the smallest of the most frequent values"""
    v, c = np.unique(values, return_counts=True)
    return v[np.argmax(c)]


def region_meanshift(existing_offsets, pred_embedding, pred_semantic, pred_displacements, xyz, pred_bw, regions,
                     bw_relax_factor=1.0, sem_len=None, expected_instance_num=300, max_iter=10):
    """examples/ScanNet/evaluate_instance.py:646-806 (`region_based_cross_modal_meanshift_clustering`) without its
    visualisation branches; float64.  Returns the instance id of every point (-1 = background)."""
    f = np.float64
    emb, pose = np.asarray(pred_embedding, f), np.asarray(xyz, f) - np.asarray(pred_displacements, f)
    sem_len = emb.shape[1] if sem_len is None else sem_len
    region_index, region_mask = np.unique(regions, return_inverse=True)
    n = len(region_index)
    E = np.concatenate([gaussian_weighted_region_mean(emb, region_mask, n),
                        gaussian_weighted_region_mean(pose, region_mask, n)], 1)
    BW = gaussian_weighted_region_mean(np.asarray(pred_bw, f), region_mask, n)
    center_sem = np.array([_mode(np.asarray(pred_semantic)[region_mask == i]) for i in range(n)])
    npts = S_.counts(region_mask, n)
    off = S_.scatter_max(np.asarray(existing_offsets, f).reshape(len(region_mask), -1), region_mask, n)[0][:, 0]
    background = np.ones(n, bool)
    pred = np.zeros(n, np.int32)
    flag = (center_sem < 2) & background
    off[flag] = 0
    background[flag] = False
    pred[flag] = -1
    count = 0
    while count < expected_instance_num and background.sum() > 0:
        key = int(np.argmax(off))
        pre = E[key]
        b = BW[key] * bw_relax_factor
        dist = E - pre
        prob = _cross_modal_gaussian(dist[:, :sem_len], dist[:, sem_len:sem_len + 3], b[0], b[1])
        flag = (prob > 0.5) & background
        w = _gaussian(np.linalg.norm(dist, axis=1), b.mean())[flag] * npts[flag]
        pre = _weighted_mean(E[flag], w)
        for _ in range(max_iter):
            dist = E - pre
            b = BW[flag].mean(0) * bw_relax_factor
            prob = _cross_modal_gaussian(dist[:, :sem_len], dist[:, sem_len:sem_len + 3], b[0], b[1])
            flag = (prob > 0.5) & background
            w = _gaussian(np.linalg.norm(dist, axis=1), b.mean())[flag] * npts[flag]
            centroid = _weighted_mean(E[flag], w)
            shift = pre - centroid
            pre = centroid
            if np.linalg.norm(shift) < 1e-5:
                break
        label = _mode(center_sem[flag])
        flag = flag & (center_sem == label)
        background[flag] = False
        off[flag] = 0
        count += 1
        pred[flag] = count if flag.sum() > 0 else -1
    return pred[region_mask]
