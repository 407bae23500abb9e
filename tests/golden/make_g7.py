#!/usr/bin/env python3
"""Generate tests/golden/g7_trainmerge.npz.  BUILD CONTAINER ONLY: needs /root/reference.

Imports the reference's examples/ScanNet/datasets/scannet.py UNCHANGED, builds its ``ScanNet`` object over three small
synthetic ``*_instance.pth`` scenes (written to a temp directory in the format of examples/ScanNet/prepare_data.py:95-112),
and stores what ITS ``trainMerge`` and ``valMerge`` return for them -- with elastic distortion, rotation noise, normals
and the 'lcnh' features switched on -- together with the scenes themselves.  tests/test_data_collate.py feeds the same
scenes and the same seeds (ScanNet.__init__ seeds numpy and torch with 100, scannet.py:42-45) to
occuseg_amd.data.ScanNetCollate and compares every key.
The one third-party import the file needs that this image lacks, torch_scatter (scatter_add on CPU tensors,
scannet.py:173), is provided by a 6-line index_add_ function; nothing else is stubbed."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/examples/ScanNet/datasets/scannet.py"
CFG = {"scale": 50, "val_reps": 1, "batch_size": 2, "dimension": 3, "full_scale": 4096, "use_normal": True,
       "use_elastic": True, "use_feature": "lcnh", "use_rotation_noise": True, "regress_sigma": 0.2}


def scene(seed, P):
    rng = np.random.default_rng(seed)
    n = rng.normal(size=(P, 3))
    inst = rng.choice([3, 7, 8, 100, 250, 251], P)
    return {"coords": (rng.random((P, 3)) * [4.0, 3.0, 2.5]).astype(np.float32),
            "colors": rng.uniform(-1, 1, (P, 3)).astype(np.float32),
            "w": np.stack([rng.integers(0, 20, P), inst], 1).astype(np.float64),
            "normals": (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32),
            "region": (inst * 10 + rng.integers(0, 4, P)).astype(np.int64)}


def load_reference():
    ts = types.ModuleType("torch_scatter")

    def scatter_add(src, index, dim=0, dim_size=None):
        n = int(index.max()) + 1 if dim_size is None else dim_size
        return torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
    ts.scatter_add = scatter_add
    ts.scatter_mean = ts.scatter_std = ts.scatter_max = None  # imported by name, never called by the collate functions
    sys.modules.setdefault("torch_scatter", ts)
    import scipy.interpolate  # noqa: F401  (the file says scipy.interpolate.* after importing scipy.ndimage only)
    spec = importlib.util.spec_from_file_location("occuseg_reference_scannet", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def flat(prefix, b, out):
    locs, feats, normals, bs = b["x"]
    out[prefix + "locs"], out[prefix + "feats"], out[prefix + "normals"] = locs.numpy(), feats.numpy(), normals.numpy()
    out[prefix + "batch"] = np.int64(bs)
    for k in ("y", "instance_masks", "instance_sizes", "sizes", "offsets", "displacements", "regions", "region_masks",
              "region_indexs"):
        out[prefix + k] = b[k].numpy()
    out[prefix + "mask_argmax"] = np.concatenate([m.argmax(1).numpy() for m in b["masks"]])
    out[prefix + "mask_rowsum"] = np.concatenate([m.sum(1).numpy() for m in b["masks"]])
    out[prefix + "idxs"] = np.concatenate([i.numpy() for i in b["idxs"]])
    if "point_ids" in b:
        out[prefix + "point_ids"] = b["point_ids"].numpy()


def main():
    ref = load_reference()
    scenes = [scene(70 + i, 1500 + 200 * i) for i in range(3)]
    tmp = tempfile.mkdtemp(prefix="g7_")
    for i, s in enumerate(scenes):
        torch.save(s, os.path.join(tmp, "scene%04d_00_vh_clean_2_instance.pth" % i))
    ds = ref.ScanNet(os.path.join(tmp, "*_instance.pth"), os.path.join(tmp, "*_instance.pth"), CFG)  # seeds 100
    out = {}
    for i, s in enumerate(scenes):
        for k, v in s.items():
            out["scene%d_%s" % (i, k)] = v
    flat("train_", ds.trainMerge([2, 0], scenes), out)
    val_offsets = [0, 1500, 3200, 5100]
    flat("val_", ds.valMerge([1, 2], scenes, val_offsets), out)
    out["arg_valOffsets"] = np.array(val_offsets)
    path = os.path.join(ROOT, "tests", "golden", "g7_trainmerge.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
