"""upsample_feature / add_feature_planes of the fork (sparseconvnet/utils.py:72-140): features of a low-resolution
tensor scattered to the sites of a high-resolution one through ``SCN.ResolutionBasedScattering``.

Same results as the reference's per-sample Python loop, computed for the whole batch at once: the low-resolution
tensor's own Metadata already holds the sorted site keys of its scale, so nothing is re-hashed per call."""
import torch

from .. import SCN
from .._lib import check, load, ptr, stream
from .tensor import SparseConvNetTensor


def _lookup(lr, coords_i32, div):
    """global low-resolution site id of every row of coords_i32 [n,4] (x,y,z,sample) divided by div, or -1"""
    g = lr.metadata.grids[SCN._size_key(lr.spatial_size)]
    out = torch.empty(coords_i32.size(0), dtype=torch.int32, device=coords_i32.device)
    check(load().scn_lookup_sites(ptr(g.keys), g.n, ptr(coords_i32), 4, coords_i32.size(0), int(div), 0, ptr(out),
                                  stream()))
    return out


def upsample_feature(lr, hr, stride, bilinear=False):
    dev = lr.features.device
    g_lr = lr.metadata.grids[SCN._size_key(lr.spatial_size)]
    g_hr = hr.metadata.grids[SCN._size_key(hr.spatial_size)]
    loc_hr = g_hr.coords  # int32 [N,4] on the device, site order
    output = SparseConvNetTensor(metadata=hr.metadata, spatial_size=hr.spatial_size)
    # first low-resolution site of each sample: the reference adds it to the per-sample ranks
    # (sparseconvnet/utils.py:83-90,136); a point without a parent ends up at lr_start - 1, clamped at 0, exactly as there
    batch = int(loc_hr[:, 3].max().item()) + 1 if loc_hr.numel() else 0
    lr_start = torch.zeros(batch + 1, dtype=torch.int64, device=dev)
    if g_lr.n:
        lr_start[1:] = torch.cumsum(torch.bincount(g_lr.coords[:, 3].long(), minlength=batch)[:batch], 0)
    missing = (lr_start[loc_hr[:, 3].long()] - 1).clamp_(min=0)
    if not bilinear:
        idx = _lookup(lr, loc_hr.contiguous(), stride).long()
        idx = torch.where(idx >= 0, idx, missing)
        output.features = torch.index_select(lr.features, 0, idx)
        return output
    # bilinear: the 8 low-resolution cells around (x - (stride-1)/2) / stride, weights as sparseconvnet/utils.py:94-130
    cand = (loc_hr[:, :3].float() - (stride - 1) / 2) / stride
    ceil_c, floor_c = torch.ceil(cand), torch.floor(cand)
    diff = [ceil_c - cand, cand - floor_c]
    anchors = [ceil_c, floor_c]
    feats = torch.zeros((loc_hr.size(0), lr.features.size(1)), dtype=lr.features.dtype, device=dev)
    total = torch.zeros(loc_hr.size(0), dtype=lr.features.dtype, device=dev)
    for x in (0, 1):
        for y in (0, 1):
            for z in (0, 1):
                w = (1 - diff[x][:, 0]) * (1 - diff[y][:, 1]) * (1 - diff[z][:, 2])
                q = loc_hr.clone()
                q[:, 0] = anchors[x][:, 0].int()
                q[:, 1] = anchors[y][:, 1].int()
                q[:, 2] = anchors[z][:, 2].int()
                idx = _lookup(lr, q.contiguous(), 1).long()
                w = torch.where(idx >= 0, w, torch.zeros_like(w))
                idx = torch.where(idx >= 0, idx, lr_start[loc_hr[:, 3].long()])
                feats = feats + torch.index_select(lr.features, 0, idx) * w.view(-1, 1)
                total = total + w
    output.features = feats / total.view(-1, 1)
    return output


def add_feature_planes(input):
    output = SparseConvNetTensor(metadata=input[0].metadata, spatial_size=input[0].spatial_size)
    output.features = sum(i.features for i in input)
    return output
