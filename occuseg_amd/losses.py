"""The per-instance part of the reference's loss, without its Python loops.

examples/ScanNet/train_instance.py:204-242 (``calculate_cost``) walks the scenes of a batch and, inside, every instance
of a scene -- ``for mid in range(mask_size): cls = pred_semantics[instance_indices][0]; if cls > 1: ...`` -- with one
boolean mask over all points and one host synchronisation per instance (a few hundred per step, after a backbone that
now takes tens of milliseconds).  ``instance_regression_losses`` computes the same two numbers (DisplacementLoss,
OccupancyLoss) for the whole batch with segment reductions on the device (occuseg_amd.segment = torch_scatter's
scatter_mean / scatter_std over HIP segment kernels): no loop over scenes or instances, no ``.item()``, differentiable.

Per scene c and instance m (ids 0..n_c-1 as the collate function numbers them, examples/ScanNet/datasets/scannet.py:
180-200), with the points of the instance in batch order:
    cls(c, m)   = semantic label of the instance's FIRST point                      (train_instance.py:232-233)
    d_err(c, m) = mean_p |displacement_p - displacement_gt_p|_2                       (:224)
    o_err(c, m) = mean_p |occupancy_p - occupancy_gt_p|_2 + std_p(occupancy_p)       (:228-229,236; unbiased std)
    DisplacementLoss = 1/B sum_c [n_valid_c > 0] (sum_{m: cls > 1} d_err) / n_valid_c   (:235-243,252-254)
    OccupancyLoss    = the same with o_err
where B = number of scenes and n_valid_c = |{m: cls(c, m) > 1}|.
"""
import torch

from . import segment


def instance_segments(batch_index, instance_mask, sizes):
    """global segment id of every point: instances of scene c are numbered offset_c + 0..sizes[c]-1.
    ``sizes``: instances per scene, a HOST sequence (the collate function's batch['sizes']): no device round trip."""
    sizes = [int(s) for s in sizes]
    offsets = [0]
    for s in sizes:
        offsets.append(offsets[-1] + s)
    dev = instance_mask.device
    off = torch.tensor(offsets[:-1], dtype=torch.int64, device=dev)
    scene_of_segment = torch.repeat_interleave(torch.arange(len(sizes), device=dev),
                                               torch.tensor(sizes, dtype=torch.int64, device=dev), output_size=offsets[-1])
    seg = off[batch_index.to(torch.int64)] + instance_mask.to(torch.int64)
    return seg, scene_of_segment, offsets[-1]


def instance_regression_losses(displacements, displacements_gt, occupancy, occupancy_gt, semantics, batch_index,
                               instance_mask, sizes, foreground_above=1):
    """returns (DisplacementLoss, OccupancyLoss), each a 0-d tensor = the reference's value after ``/= batchSize``.
    displacements [P,3], occupancy [P,1], *_gt alike, semantics int [P] (batch['y'][:,0]), batch_index int [P] (column
    `dimension` of batch['x'][0]), instance_mask int [P] (batch['instance_masks']), sizes: instances per scene (host)."""
    P = displacements.size(0)
    seg, scene_of_segment, S = instance_segments(batch_index, instance_mask, sizes)
    B = len(sizes)
    dev = displacements.device
    if S == 0:
        z = torch.zeros((), dtype=torch.float32, device=dev)
        return z, z.clone()
    d_point = torch.norm(displacements - displacements_gt, dim=1)
    o_point = torch.norm(occupancy.view(P, -1) - occupancy_gt.view(P, -1), dim=1)
    d_err = segment.scatter_mean(d_point, seg, dim=0, dim_size=S)
    o_err = segment.scatter_mean(o_point, seg, dim=0, dim_size=S) + \
        segment.scatter_std(occupancy.view(P, -1), seg, dim=0, dim_size=S).sum(1)
    # class of the first point of every instance: smallest point index of the segment, then one gather
    first = torch.full((S,), P, dtype=torch.int64, device=dev)
    first.scatter_reduce_(0, seg, torch.arange(P, device=dev), "amin")
    present = first < P
    cls = semantics.to(torch.int64)[first.clamp(max=max(P - 1, 0))]
    valid = (present & (cls > foreground_above)).to(torch.float32)
    n_valid = torch.zeros(B, dtype=torch.float32, device=dev).index_add_(0, scene_of_segment, valid)
    scale = torch.where(n_valid > 0, 1.0 / n_valid.clamp(min=1), torch.zeros_like(n_valid)) / B
    w = valid * scale[scene_of_segment]
    return (d_err * w).sum(), (o_err * w).sum()
